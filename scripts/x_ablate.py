#!/usr/bin/env python
"""What bounds the hand-scheduled contraction kernels (conv_gemm_x.h)?  Times a few shapes under the production
library and under the -DAA_X_ABLATE variants (no DMA / no MFMA / no fragment reads; results are garbage, only
the time matters) - and the compiled 8-wave kernel with its runtime ablation bits for comparison."""
import os
import os as _os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animate_anything_amd import _lib, ops, build as _build  # noqa: E402

DT, dev = torch.float16, "cuda"
ops.AUTOTUNE = False
B, T = 2, 17
N = B * T


def rnd(*s):
    return torch.randn(*s, device=dev, dtype=torch.float32).to(DT)


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def shapes():
    out = []
    x = rnd(N * 64 * 64, 320)
    pw = ops.pack_weight(rnd(320, 320, 3, 3) * 0.02, rnd(320))
    g = ops.conv3x3_geom(N, 64, 64)
    out.append(("conv3x3 L0 320->320", lambda: ops.conv_gemm(x, pw, g), 2.0 * g.rows * 9 * 320 * 320, (14, 37, 39)))
    x1 = rnd(N * 16 * 16, 1280)
    pw1 = ops.pack_weight(rnd(1280, 1280, 3, 3) * 0.02, rnd(1280))
    g1 = ops.conv3x3_geom(N, 16, 16)
    out.append(("conv3x3 L2 1280->1280", lambda: ops.conv_gemm(x1, pw1, g1), 2.0 * g1.rows * 9 * 1280 * 1280, (15, 36, 38, 40)))
    M = N * 64 * 64
    x2 = rnd(M, 1280)
    pw2 = ops.pack_weight(rnd(320, 1280) * 0.05, rnd(320))
    out.append(("linear ff2 L0 K=1280 N=320", lambda: ops.conv_gemm(x2, pw2, ops.linear_geom(M)), 2.0 * M * 1280 * 320, (14, 37, 39)))
    x3 = rnd(M, 320)
    pw3 = ops.pack_weight(rnd(2560, 320) * 0.05, rnd(2560), geglu=True)
    out.append(("linear geglu L0 K=320", lambda: ops.conv_gemm(x3, pw3, ops.linear_geom(M)), 2.0 * M * 320 * 2560, (15, 36, 40)))
    M2 = N * 16 * 16
    x4 = rnd(M2, 5120)
    pw4 = ops.pack_weight(rnd(1280, 5120) * 0.05, rnd(1280))
    out.append(("linear ff2 L2 K=5120 N=1280", lambda: ops.conv_gemm(x4, pw4, ops.linear_geom(M2)), 2.0 * M2 * 5120 * 1280, (15, 36, 38, 40)))
    return out


SH = shapes()
if _os.environ.get('X_ONLY'):
    SH = [t for t in SH if _os.environ['X_ONLY'] in t[0]]
if _os.environ.get('X_CFGS'):
    _c = tuple(int(x) for x in _os.environ['X_CFGS'].split(','))
    SH = [(n, f, fl, _c) for n, f, fl, _ in SH]
ABLS = [int(x) for x in _os.environ.get('X_ABLS', '0,1,2,4,5,6,7').split(',')]
for abl in ABLS:
    lib = _lib.bind(_build.build(ablate=abl)) if abl else _lib.get()
    with _lib.use_library(lib):
        for name, fn, flops, cfgs in SH:
            for cfg in cfgs:
                is_x = cfg >= 36
                if not is_x and abl not in (0, 1, 2, 3) and abl < 8:
                    continue
                lib.aa_set_tile_override(cfg)
                ops.DEBUG_ABLATE = 0 if is_x else (abl & 3)
                us = timeit(fn)
                ops.DEBUG_ABLATE = 0
                lib.aa_set_tile_override(-1)
                what = {0: "full", 1: "no DMA", 2: "no MFMA", 4: "no reads", 5: "no DMA, no reads", 6: "no MFMA, no reads", 7: "nothing", 8: "no GELU", 16: "no stores", 15: "nothing, no GELU", 32: "no barrier", 64: "no DMA wait", 96: "no barrier, no DMA wait", 23: "nothing, no stores", 31: "nothing, no GELU, no stores"}.get(abl, str(abl))
                print(f"{name:30s} cfg {cfg:2d} {what:18s} {us:9.1f} us  {flops / us / 1e6:7.1f} TF-equiv", flush=True)
