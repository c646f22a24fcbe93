#!/usr/bin/env python
"""A/B partner of scripts/probe/gemm_8phase.hip: the SAME pure GEMMs (C = A W^T, uniform [-1, 1) operands) through the library's
contraction kernels (aa_conv_gemm as a 1x1 / linear call) under forced tile shapes - what the product's own K loops reach when the
im2col addressing is a no-op.  Usage: python scripts/gemm_library_ab.py [--tiles 36,38,40,41,43,46] [--dtype fp16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animate_anything_amd import ops, _lib  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--tiles", default="1,21,36,38,40,41,42,43,46,48")
p.add_argument("--dtype", default="fp16")
p.add_argument("--shapes", default="4096x4096x4096,8192x8192x8192,8704x1280x11520,34816x512x5760")
a = p.parse_args()
DT = torch.float16 if a.dtype == "fp16" else torch.bfloat16
ops.AUTOTUNE = False
lib = _lib.get()


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return best, e0.elapsed_time(e1) * 1e-3 / reps


for shape in a.shapes.split(","):
    M, N, K = (int(v) for v in shape.split("x"))
    x = (torch.rand(M, K, device="cuda") * 2 - 1).to(DT)
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(DT)
    pw = ops.pack_weight(w, None)
    g = ops.linear_geom(M)
    ref = None
    for t in [-1] + [int(v) for v in a.tiles.split(",")]:
        if t >= 0:
            bm, bn, bk, st = ops.TILE_TABLE[t]
            if pw.n_pad % bn:
                continue
        lib.aa_set_tile_override(t)
        out = ops.conv_gemm(x, pw, g)
        if ref is None:
            ref = (x[:64].float() @ w.float().t())
        err = (out[:64].float() - ref).abs().max().item() / ref.abs().max().item()
        best, mean = timeit(lambda: ops.conv_gemm(x, pw, g))
        fl = 2.0 * M * N * K
        name = "auto" if t < 0 else f"tile {t}: {ops.TILE_TABLE[t]}"
        print(f"M={M:6d} N={N:6d} K={K:6d}  {name:34s} rel_err {err:.1e}  best {best * 1e6:8.1f} us = {fl / best / 1e12:7.1f} TF/s   back-to-back {fl / mean / 1e12:7.1f} TF/s", flush=True)
    lib.aa_set_tile_override(-1)
