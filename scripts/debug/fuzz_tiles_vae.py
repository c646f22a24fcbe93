"""Tile fuzzing of the AutoencoderKL (SD configuration 128/256/512/512) by SELF-CONSISTENCY (r04): decode of four 64x64 latents and
encode of two 512x512 frames with a random eligible (tile, K splits) pair per contraction signature, against the same call on the
autotuner-free library heuristic.  Different tiles round alike (fp32 accumulation, one rounding): a difference beyond 1.5e-2 of
the output range marks a wrong tile.  Usage: fuzz_tiles_vae.py [iterations] [seed]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from animate_anything_amd import ops  # noqa: E402
from animate_anything_amd.vae import AutoencoderKL  # noqa: E402

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.manual_seed(0)
vae = AutoencoderKL().eval()
with torch.no_grad():
    for p_ in vae.parameters():
        if p_.abs().max() == 0:
            p_.normal_(0.0, 0.02)
vae = vae.half().cuda()
g = torch.Generator().manual_seed(5)
z = torch.randn(4, 4, 64, 64, generator=g).half().cuda()
img = (torch.rand(2, 3, 512, 512, generator=g) * 2 - 1).half().cuda()


fixed, log = {}, []


def pick(key, cands):
    if key not in fixed:
        fixed[key] = rng.choice(cands)
        log.append((key, fixed[key]))
    return fixed[key]


def run():
    with torch.no_grad():
        return vae.decode(z).sample.float(), vae.encode(img).latent_dist.mode().float()


ops.AUTOTUNE = False
ref_dec, ref_enc = run()                                     # library heuristic, no autotuning
torch.cuda.synchronize()
ops.TILE_PICKER = pick
fails = 0
for it in range(n_it):
    fixed.clear()
    log.clear()
    dec, enc = run()
    torch.cuda.synchronize()
    ed = ((dec - ref_dec).abs().max() / ref_dec.abs().max()).item()
    ee = ((enc - ref_enc).abs().max() / ref_enc.abs().max()).item()
    bad = not (ed < 1.5e-2 and ee < 1.5e-2 and torch.isfinite(dec).all() and torch.isfinite(enc).all())
    print(f"vae {it}: decode diff {ed:.4f} encode diff {ee:.4f} {'FAIL' if bad else 'ok'}", flush=True)
    if bad:
        fails += 1
        for k, c in log:
            print("   ", k, c)
print("failures:", fails)
