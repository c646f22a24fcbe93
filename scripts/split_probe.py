#!/usr/bin/env python
"""Time one contraction under explicit (tile, K splits) choices.  Usage: split_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animate_anything_amd import _lib, ops  # noqa: E402

ops.AUTOTUNE = False
lib = _lib.get()
DT = torch.float16


def rnd(*s):
    return torch.randn(*s, device="cuda", dtype=torch.float32).to(DT)


def run(name, fn, flops, combos):
    for tile, sp in combos:
        lib.aa_set_tile_override(tile)
        ops.K_SPLITS = sp
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = sorted(ts)[2] * 1e-3
        print(f"{name:34s} tile {tile:2d} splits {sp}: {t * 1e6:8.1f} us  {flops / t / 1e12:7.1f} TFLOP/s", flush=True)
    lib.aa_set_tile_override(-1)
    ops.K_SPLITS = 0


n = 34
x = rnd(n * 16 * 16, 1280)
w = ops.pack_weight(rnd(1280, 1280, 3, 3), rnd(1280))
g = ops.conv3x3_geom(n, 16, 16)
run("conv3x3 1280->1280 @16x16", lambda: ops.conv_gemm(x, w, g), 2.0 * g.rows * 11520 * 1280,
    [(15, 0), (15, 3), (12, 0), (2, 0), (23, 0), (24, 0), (25, 0), (16, 0)])
w2 = ops.pack_weight(rnd(1280, 1280), rnd(1280))
run("linear 1280->1280 M=8704", lambda: ops.conv_gemm(x, w2, ops.linear_geom(n * 256)), 2.0 * n * 256 * 1280 * 1280,
    [(3, 0), (16, 0), (12, 0), (23, 0), (24, 0), (25, 0)])
w3 = ops.pack_weight(rnd(1280, 1280, 3, 1, 1), rnd(1280))
gt = ops.tconv_geom(2, 17, 256)
run("tconv 1280 M=8704 K=3840", lambda: ops.conv_gemm(x, w3, gt), 2.0 * gt.rows * 3840 * 1280, [(15, 0), (12, 0), (16, 0), (23, 0), (24, 0), (25, 0)])
