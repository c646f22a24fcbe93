#!/usr/bin/env python
"""Where does a workgroup of the hand-scheduled contraction kernels (conv_gemm_x.h) spend its cycles, and how long does a CU
sit between two workgroups?  AaConvGemm.debug bit 8: thread 0 of every workgroup stamps the shader clock at
entry | before the first DMA piece | first stage landed | behind the K loop | exit, the 100 MHz wall clock at entry and exit,
and the CU it ran on.  Prints mean cycles per phase and, per CU, the busy fraction and the mean gap between the exit of one
workgroup and the entry of the next."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animate_anything_amd import _lib, ops  # noqa: E402

ONLY = os.environ.get("PROBE_ONLY", "")
DT, dev = torch.float16, "cuda"
ops.AUTOTUNE = False
lib = _lib.get()


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
        ops.DEBUG_ABLATE = 8
        fn(); torch.cuda.synchronize()
        ops.DEBUG_ABLATE = 0
        st = ops.LAST_STAMPS[0]
        st = st[st[:, 5] != 0]
        d = st.double()
        ph = [(d[:, b] - d[:, a]).mean().item() for a, b in ((0, 1), (1, 6), (6, 2), (2, 5))]
        tot = (d[:, 5] - d[:, 0]).mean().item()
        # wall-clock view (10 ns ticks): per CU (xcc, se, sh, cu) sort by entry, gaps between exit and next entry
        hw = st[:, 3]
        cu_key = ((hw >> 32) & 0xF) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xF)
        t_in, t_out = st[:, 7], st[:, 4]
        span = (t_out.max() - t_in.min()).item() * 10.0
        keys = cu_key.unique()
        busy, gaps, first, per_cu = [], [], [], []
        t0 = t_in.min().item()
        for k in keys.tolist():
            sel = cu_key == k
            a, b = t_in[sel], t_out[sel]
            o = a.argsort()
            a, b = a[o], b[o]
            per_cu.append(a.numel())
            first.append((a[0].item() - t0) * 10.0)
            # co-resident workgroups overlap: busy = union of intervals
            cur_a, cur_b, tot_b = a[0].item(), b[0].item(), 0
            for x, y in zip(a[1:].tolist(), b[1:].tolist()):
                if x > cur_b:
                    gaps.append((x - cur_b) * 10.0)
                    tot_b += cur_b - cur_a
                    cur_a, cur_b = x, y
                else:
                    cur_b = max(cur_b, y)
            tot_b += cur_b - cur_a
            busy.append(tot_b * 10.0)
        import statistics as S
        print(f"{name:34s} [{cfg}] wall {wall:6.1f} us span {span / 1e3:6.1f} us wgs {st.shape[0]:5d} on {len(keys)} CUs ({min(per_cu)}-{max(per_cu)} each) | "
              f"clk: setup {ph[0]:6.0f} fill {ph[1]:6.0f} kloop {ph[2]:7.0f} epilogue {ph[3]:7.0f} total {tot:7.0f} | "
              f"CU busy {S.mean(busy) / 1e3:6.1f} us, first entry +{S.mean(first) / 1e3:5.2f} us, gap between workgroups "
              f"{(S.mean(gaps) if gaps else 0):6.0f} ns x {len(gaps) / max(1, len(keys)):4.1f}", flush=True)
    lib.aa_set_tile_override(-1)


M = 34 * 64 * 64
x320 = rnd(M, 320)
res = rnd(M, 320)
x1280 = rnd(M, 1280)
wg = ops.pack_weight(rnd(2560, 320) * 0.05, rnd(2560), geglu=True)
probe("geglu K=320 N=2560", lambda: ops.conv_gemm(x320, wg, ops.linear_geom(M)), [36, 40, 41, 43, 44])
w = ops.pack_weight(rnd(960, 320) * 0.05)
probe("linear qkv K=320 N=960", lambda: ops.conv_gemm(x320, w, ops.linear_geom(M)), [37, 39, 45])
w = ops.pack_weight(rnd(320, 320) * 0.05, rnd(320))
probe("linear K=320 N=320 +res", lambda: ops.conv_gemm(x320, w, ops.linear_geom(M), residual=res), [37, 39, 45])
w = ops.pack_weight(rnd(320, 1280) * 0.05, rnd(320))
probe("linear ff2 K=1280 N=320 +res", lambda: ops.conv_gemm(x1280, w, ops.linear_geom(M), residual=res), [37, 39])
w = ops.pack_weight(rnd(320, 320, 3, 3) * 0.02, rnd(320))
probe("conv3x3 320->320", lambda: ops.conv_gemm(x320, w, ops.conv3x3_geom(34, 64, 64)), [37, 39])
M2 = 34 * 32 * 32
x640 = rnd(M2, 640)
wg = ops.pack_weight(rnd(5120, 640) * 0.05, rnd(5120), geglu=True)
probe("geglu K=640 N=5120", lambda: ops.conv_gemm(x640, wg, ops.linear_geom(M2)), [36, 40, 44])
M3 = 34 * 16 * 16
x1 = rnd(M3, 5120)
w = ops.pack_weight(rnd(1280, 5120) * 0.05, rnd(1280))
probe("linear ff2 K=5120 N=1280 M=8704", lambda: ops.conv_gemm(x1, w, ops.linear_geom(M3)), [36, 38, 40])
