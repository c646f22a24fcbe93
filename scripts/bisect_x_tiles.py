#!/usr/bin/env python
"""Which hand-scheduled tile breaks the full-size forward?  One model, the committed tile cache, the golden of
tests/test_gpu_fullsize.py; the forward is repeated with subsets of the x tiles withdrawn (aa_set_tile_override(-100 - mask))."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from animate_anything_amd import _lib, ops  # noqa: E402
from animate_anything_amd.unet3d import UNet3DConditionModel  # noqa: E402
from util import FULL_UNET, fullsize_inputs, fullsize_oracle  # noqa: E402

lat = int(os.environ.get("LAT", "64"))
want = torch.load(os.path.join(ROOT, "tests", "golden", f"unet_fullsize_16x{lat}x{lat}.pt"))["out"].float()
_, state = fullsize_oracle()
net = UNet3DConditionModel(**FULL_UNET).eval()
net.load_state_dict(state)
del state
net = net.half().cuda()
i = fullsize_inputs(16, lat)
dev = lambda x: x.half().cuda()
lib = _lib.get()
NX = 10
masks = [("all x tiles offered", 0), ("no x tiles", (1 << NX) - 1)] + [(f"only x tile {36 + k}", ((1 << NX) - 1) & ~(1 << k)) for k in range(NX)]
for name, m in masks:
    lib.aa_set_tile_override(-100 - m)
    with torch.no_grad():
        got = net(dev(i["sample"]), i["t"], dev(i["text"]), dev(i["cond"]), dev(i["mask"]), motion=i["motion"]).sample.float().cpu()
    mse = ((got - want) ** 2).mean().item()
    print(f"{name:24s} mse {mse:.3e}  autotuned so far {ops.AUTOTUNE_EVENTS}", flush=True)
lib.aa_set_tile_override(-100)
