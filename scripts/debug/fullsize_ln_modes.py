"""Debug aid (r04i): the full-size forward of tests/test_gpu_fullsize.py under the LayerNorm-fold variants, same process = same
autotuned tiles.  Prints the max-normalised error / MSE against the oracle golden per variant and where the worst element sits."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from animate_anything_amd import layers, ops  # noqa: E402
from animate_anything_amd.unet3d import UNet3DConditionModel  # noqa: E402
from util import FULL_UNET, fullsize_inputs, fullsize_oracle, rel_err  # noqa: E402

DT = torch.float16
want = torch.load(os.path.join(ROOT, "tests", "golden", "unet_fullsize_16x64x64.pt"))["out"].float()
_, state = fullsize_oracle()
i = fullsize_inputs(16, 64)
net = UNet3DConditionModel(**FULL_UNET).eval()
net.load_state_dict(state)
del state
net = net.to(DT).cuda()
dev = lambda x: x.to(DT).cuda()


def run(tag):
    with torch.no_grad():
        got = net(dev(i["sample"]), i["t"], dev(i["text"]), dev(i["cond"]), dev(i["mask"]), motion=i["motion"]).sample
    torch.cuda.synchronize()
    got = got.float().cpu()
    d = (got - want).abs()
    idx = torch.nonzero(d == d.max())[0].tolist()
    print(f"{tag:28s} rel_err {rel_err(got, want):.4f}  mse {((got - want) ** 2).mean().item():.3g}  worst at {idx}  "
          f"elements > 0.03*max: {(d > 0.03 * want.abs().max()).sum().item()}", flush=True)
    return got


a = run("raw 2/10 parts (default)")
a2 = run("raw again")
print("bit-equal run to run:", torch.equal(a, a2))
ops.LN_FINALIZE_LAUNCH = True
b = run("aa_ln_finalize for all")
ops.LN_FINALIZE_LAUNCH = False
ops.LN_RAW_ANY_PARTS = True
c = run("raw for any part count")
ops.LN_RAW_ANY_PARTS = False
layers.LN_FOLD = False
e = run("no fold (LayerNorm kernels)")
layers.LN_FOLD = True
print("raw vs finalize max diff", (a - b).abs().max().item(), " raw-any vs finalize", (c - b).abs().max().item(), " nofold vs finalize", (e - b).abs().max().item())
for k, v in sorted(ops._tile_cache.items()):
    if "ln" in k or "stats" in k:
        print(k, v)
