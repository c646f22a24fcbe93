"""Tile fuzzing at the metric configuration (r04): every aa_conv_gemm call of the full-size UNet3D forward runs with a RANDOM eligible
(tile, K splits) pair instead of the autotuner's choice; the forward must match the oracle golden for every assignment.  Eager
iterations draw per call; graph iterations draw one assignment per signature, capture, replay twice.  A failing iteration prints its
assignment (signature -> choice) so that the offending tile can be bisected.  Usage: fuzz_tiles_fullsize.py [eager iters] [graph iters] [seed] [fp16|bf16|rgba|svd] [probability of an arbitrary K split]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from animate_anything_amd import ops  # noqa: E402
from animate_anything_amd.unet3d import UNet3DConditionModel  # noqa: E402
from util import FULL_UNET, fullsize_inputs, fullsize_oracle, rel_err  # noqa: E402

n_eager = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_graph = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = random.Random(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
MODE = sys.argv[4] if len(sys.argv) > 4 else "fp16"          # fp16 | bf16 (the UNet3D step) | rgba (configs[4]: 48x48 latents) | svd (configs[3], fp16)
DT = torch.bfloat16 if MODE == "bf16" else torch.float16
TOL_E, TOL_M = (1.5e-1, 1e-2) if MODE == "bf16" else (3e-2, 1e-3)
dev = lambda x: x.to(DT).cuda()
if MODE == "svd":
    from animate_anything_amd.svd_unet import UNetSpatioTemporalConditionModel
    from util import FULL_SVD_UNET, fullsize_svd_oracle, svd_unet_inputs
    want = torch.load(os.path.join(ROOT, "tests", "golden", "svd_unet_fullsize_14x72x128.pt"))["out"].float()
    TOL_M = 1e-3 * max((want ** 2).mean().item(), 1.0)
    _, state = fullsize_svd_oracle()
    i = svd_unet_inputs(2, 14, 72, 128)
    net = UNetSpatioTemporalConditionModel(**FULL_SVD_UNET).eval()
    net.load_state_dict(state)
    del state
    net = net.to(DT).cuda()
    args, kwargs = (dev(i["sample"]), i["t"], dev(i["text"]), i["ids"].cuda()), {}
else:
    lat = 48 if MODE == "rgba" else 64
    want = torch.load(os.path.join(ROOT, "tests", "golden", f"unet_fullsize_16x{lat}x{lat}.pt"))["out"].float()
    _, state = fullsize_oracle()
    i = fullsize_inputs(16, lat)
    net = UNet3DConditionModel(**FULL_UNET).eval()
    net.load_state_dict(state)
    del state
    net = net.to(DT).cuda()
    args, kwargs = (dev(i["sample"]), i["t"], dev(i["text"]), dev(i["cond"]), dev(i["mask"])), dict(motion=i["motion"])


log = []
fixed = {}


SPLIT_P = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0    # probability of an arbitrary K split count on top of the drawn tile


def pick(key, cands):
    if key not in fixed:                                     # one draw per signature and iteration
        tile, sp = rng.choice(cands)
        if SPLIT_P and rng.random() < SPLIT_P:               # (the library clamps the count to the K steps and ignores it where a call cannot split)
            sp = rng.choice((2, 3, 4, 7))
        fixed[key] = (tile, sp)
        log.append((key, fixed[key]))
    return fixed[key]


def check(tag, got):
    got = got.float().cpu()
    e, m = rel_err(got, want), ((got - want) ** 2).mean().item()
    bad = not (e < TOL_E and m < TOL_M) or not torch.isfinite(got).all()
    print(f"{tag}: rel_err {e:.4f} mse {m:.3g} {'FAIL' if bad else 'ok'}", flush=True)
    return bad


ops.TILE_PICKER = pick
fails = 0
with torch.no_grad():
    for it in range(n_eager):
        log.clear()
        fixed.clear()
        got = net(*args, **kwargs).sample
        torch.cuda.synchronize()
        fails += check(f"eager {it}", got)
        for k, c in log:                                     # (every iteration: passing ones clear a choice)
            print("   ", k, c)
    # graph mode: one assignment per signature, drawn in an eager pass, then captured
    for it in range(n_graph):
        log.clear()
        fixed.clear()
        ops.TILE_PICKER = pick
        net(*args, **kwargs)                                 # eager pass draws (and logs) a choice per signature
        assign = dict(fixed)
        ops.TILE_PICKER = lambda key, cands, assign=assign: assign.get(key, cands[0])      # the capture replays the same assignment
        net.enable_graph()
        for rep in range(2):
            got = net(*args, **kwargs).sample
        torch.cuda.synchronize()
        if check(f"graph {it}", got):
            fails += 1
            for k, c in sorted(assign.items(), key=str):
                print("   ", k, c)
        net.enable_graph(False)
        ops.TILE_PICKER = pick
print("failures:", fails)
