#!/usr/bin/env python
"""Where does a workgroup of the LDS-DMA contraction kernel spend its cycles?  AaConvGemm.debug bit 8 makes
thread 0 of every workgroup stamp the shader clock at: entry | pipeline start | end of K loop | pass-0 staged |
pass-0 stored | exit.  Prints the mean cycles of each phase per (shape, tile)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animate_anything_amd import _lib, ops  # noqa: E402

DT = torch.float16
dev = "cuda"
ops.AUTOTUNE = False
lib = _lib.get()


def rnd(*s):
    return torch.randn(*s, device=dev, dtype=torch.float32).to(DT)


def probe(name, fn, cfgs):
    for cfg in cfgs:
        lib.aa_set_tile_override(cfg)
        ops.DEBUG_ABLATE = 0
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        wall = e0.elapsed_time(e1) * 1e3
        ops.DEBUG_ABLATE = 8
        fn(); torch.cuda.synchronize()
        st = ops.LAST_STAMPS
        st = st[st[:, 5] != 0].double()
        d = (st[:, 1:6] - st[:, 0:5]).mean(0).tolist()
        tot = (st[:, 5] - st[:, 0]).mean().item()
        span = (st[:, 5].max() - st[:, 0].min()).item()
        bm, bn, bk, stg = ops.TILE_TABLE[cfg]
        print(f"{name:40s} [{cfg}:{bm}x{bn} k{bk}] wall {wall:7.1f} us  wgs {st.shape[0]:5d}  span {span:9.0f} clk | "
              f"setup {d[0]:6.0f}  kloop {d[1]:7.0f}  stage0 {d[2]:6.0f}  store0 {d[3]:6.0f}  rest {d[4]:6.0f}  total {tot:7.0f}", flush=True)
    ops.DEBUG_ABLATE = 0
    lib.aa_set_tile_override(-1)


M = 34 * 64 * 64
x320 = rnd(M, 320)
w = ops.pack_weight(rnd(960, 320), rnd(960))
probe("linear qkv K=320 N=960", lambda: ops.conv_gemm(x320, w, ops.linear_geom(M)), [4, 0, 17, 11])
w = ops.pack_weight(rnd(320, 320), rnd(320))
res = rnd(M, 320)
probe("linear C->C K=320 N=320 +res", lambda: ops.conv_gemm(x320, w, ops.linear_geom(M), residual=res), [4, 17, 11])
w = ops.pack_weight(rnd(2560, 320), rnd(2560), geglu=True)
probe("geglu K=320 N=2560", lambda: ops.conv_gemm(x320, w, ops.linear_geom(M)), [4, 14])
x1280 = rnd(M, 1280)
w = ops.pack_weight(rnd(320, 1280), rnd(320))
probe("linear ff2 K=1280 N=320 +res", lambda: ops.conv_gemm(x1280, w, ops.linear_geom(M), residual=res), [4, 11])
w = ops.pack_weight(rnd(320, 320, 3, 3), rnd(320))
probe("conv3x3 320->320", lambda: ops.conv_gemm(x320, w, ops.conv3x3_geom(34, 64, 64)), [4])
