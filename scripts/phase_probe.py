#!/usr/bin/env python
"""Where does a workgroup of the LDS-DMA contraction kernel spend its cycles?  AaConvGemm.debug bit 8 makes
thread 0 of every workgroup stamp the shader clock at: entry | pipeline start | end of K loop | first
epilogue block-row done | exit.  Prints the mean cycles of each phase per (shape, tile)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animate_anything_amd import _lib, ops  # noqa: E402

ABLATE = int(os.environ.get("PROBE_ABLATE", "0"))      # extra AaConvGemm.debug bits for the stamped launch (1: no DMA, 2: no MFMA)
ONLY = os.environ.get("PROBE_ONLY", "")
DT = torch.float16
dev = "cuda"
ops.AUTOTUNE = False
from animate_anything_amd import build as _build  # noqa: E402
lib = _lib.bind(_build.build(probe=True))          # -DAA_PHASE_PROBE variant (built here if missing)
_lib.use_library(lib).__enter__()


def rnd(*s):
    return torch.randn(*s, device=dev, dtype=torch.float32).to(DT)


def probe(name, fn, cfgs):
    if ONLY and ONLY not in name:
        return
    for cfg in cfgs:
        lib.aa_set_tile_override(cfg)
        ops.DEBUG_ABLATE = 0
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        wall = e0.elapsed_time(e1) * 1e3
        ops.DEBUG_ABLATE = 8 | ABLATE
        fn(); torch.cuda.synchronize()
        live = ops.LAST_STAMPS[0][:, 5] != 0
        kl = ops.LAST_STAMPS[1][live].double().mean(0).tolist()
        st = ops.LAST_STAMPS[0][live].double()
        d = [(st[:, b] - st[:, a]).mean().item() for a, b in ((0, 1), (1, 2), (2, 3), (3, 5))]
        tot = (st[:, 5] - st[:, 0]).mean().item()
        span = (st[:, 5].max() - st[:, 0].min()).item()
        bm, bn, bk, stg = ops.TILE_TABLE[cfg]
        print(f"{name:40s} [{cfg}:{bm}x{bn} k{bk}] wall {wall:7.1f} us  wgs {st.shape[0]:5d}  span {span:9.0f} clk | "
              f"setup {d[0]:6.0f}  kloop {d[1]:7.0f}  epi-row0 {d[2]:6.0f}  epi-rest {d[3]:6.0f}  total {tot:7.0f} | kloop: dma-wait {kl[0]:6.0f} barrier {kl[1]:6.0f} issue {kl[2]:6.0f} multiply {kl[3]:6.0f}", flush=True)
    ops.DEBUG_ABLATE = 0
    lib.aa_set_tile_override(-1)


M = 34 * 64 * 64
x320 = rnd(M, 320)
res = rnd(M, 320)
x1280 = rnd(M, 1280)
w = ops.pack_weight(rnd(320, 1280), rnd(320))
probe("linear ff2 K=1280 N=320 +res", lambda: ops.conv_gemm(x1280, w, ops.linear_geom(M), residual=res), [4, 14])
w = ops.pack_weight(rnd(320, 320, 3, 3), rnd(320))
probe("conv3x3 320->320", lambda: ops.conv_gemm(x320, w, ops.conv3x3_geom(34, 64, 64)), [4, 14, 6, 21])
M2 = 34 * 32 * 32
x640 = rnd(M2, 640)
w = ops.pack_weight(rnd(640, 640, 3, 3), rnd(640))
probe("conv3x3 640->640", lambda: ops.conv_gemm(x640, w, ops.conv3x3_geom(34, 32, 32)), [4, 14])
M3 = 34 * 16 * 16
x1 = rnd(M3, 1280)
w = ops.pack_weight(rnd(1280, 1280), rnd(1280))
probe("linear K=1280 N=1280 M=8704", lambda: ops.conv_gemm(x1, w, ops.linear_geom(M3)), [3, 15, 23, 24])
w = ops.pack_weight(rnd(1280, 1280, 3, 3), rnd(1280))
probe("conv3x3 1280 M=8704", lambda: ops.conv_gemm(x1, w, ops.conv3x3_geom(34, 16, 16)), [3, 15, 23, 24])
# short-K contractions of the first level (the epilogue-heavy ones)
wg = ops.pack_weight(rnd(2560, 320), rnd(2560), geglu=True)
probe("geglu K=320 N=2560", lambda: ops.conv_gemm(x320, wg, ops.linear_geom(M)), [3, 15, 27, 12, 33, 25])
w = ops.pack_weight(rnd(320, 320), rnd(320))
probe("linear K=320 N=320 +res", lambda: ops.conv_gemm(x320, w, ops.linear_geom(M), residual=res), [4, 14, 26, 11, 30])
w = ops.pack_weight(rnd(960, 320))
probe("linear qkv K=320 N=960", lambda: ops.conv_gemm(x320, w, ops.linear_geom(M)), [4, 14, 26, 11, 30])
