#!/usr/bin/env python
"""Per-kernel microbenchmarks at the shapes of one 16f x 512x512 denoising step:
TFLOP/s for the MFMA kernels, GB/s for the HBM-bound ones.  Random data (DVFS: zero-filled operands
overstate throughput).  Usage: python scripts/bench_kernels.py [--dtype fp16|bf16] [--only substr]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animate_anything_amd import ops  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--dtype", default="fp16")
p.add_argument("--only", default="")
p.add_argument("--reps", type=int, default=10)
p.add_argument("--ablate", type=int, default=0, help="AaConvGemm.debug ablation bits (1: no DMA, 2: no MFMA)")
p.add_argument("--cfg-sweep", action="store_true", help="time every contraction shape under every forced tile shape")
p.add_argument("--cfgs", default="", help="comma-separated tile indices the sweep is restricted to")
a = p.parse_args()
DT = torch.float16 if a.dtype == "fp16" else torch.bfloat16
dev = "cuda"


def timeit(fn, reps=a.reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def rnd(*s):
    return torch.randn(*s, device=dev, dtype=torch.float32).to(DT)


from animate_anything_amd import _lib  # noqa: E402
CFGS = list(ops.TILE_TABLE)
ONLY_CFGS = {int(c) for c in a.cfgs.split(",") if c}
ops.AUTOTUNE = False
ops.DEBUG_ABLATE = a.ablate
ops.ATTN_FLAGS = int(os.environ.get("AA_ATTN_FLAGS", "0"))


def sweep(name, fn, flops, n_pad, geglu=0):
    """fn() under the automatic tile choice and under every tile shape that divides n_pad."""
    lib = _lib.get()
    report(name + " [auto]", timeit(fn), flops=flops)
    if not a.cfg_sweep:
        return
    for i, (bm, bn, bk, st) in enumerate(CFGS):
        if ONLY_CFGS and i not in ONLY_CFGS:
            continue
        if n_pad % bn or (geglu and (bn // ops.TILE_TABLE.wave_cols(i)) % 64):
            continue
        lib.aa_set_tile_override(i)
        report(f"{name} [{i}:{bm}x{bn} k{bk} s{st}]", timeit(fn), flops=flops)
    lib.aa_set_tile_override(-1)


def report(name, secs, flops=None, bytes_=None):
    msg = f"{name:58s} {secs * 1e6:10.1f} us"
    if flops:
        msg += f"  {flops / secs / 1e12:8.1f} TFLOP/s ({flops / secs / 2.5e15 * 100:5.1f}% of 2.5PF)"
    if bytes_:
        msg += f"  {bytes_ / secs / 1e9:8.1f} GB/s ({bytes_ / secs / 8e12 * 100:5.1f}% of 8TB/s)"
    print(msg, flush=True)


B, T = 2, 17
N = B * T
LEVELS = [(320, 64), (640, 32), (1280, 16), (1280, 8)]


def want(name):
    return a.only in name


# ------------------------------------------------------------------ contraction kernel
def bench_linear(name, M, K, Nout, geglu=False):
    if not want(name):
        return
    x, w = rnd(M, K), rnd(Nout, K) * 0.05
    pw = ops.pack_weight(w, rnd(Nout), geglu=geglu)
    g = ops.linear_geom(M)
    sweep(f"{name} M={M} K={K} N={Nout}", lambda: ops.conv_gemm(x, pw, g), 2.0 * M * K * Nout, pw.n_pad, pw.geglu)


def bench_conv(name, n, hw, cin, cout, stride=1, up=False, c1=0):
    if not want(name):
        return
    x0 = rnd(n * hw * hw, cin - c1)
    x1 = rnd(n * hw * hw, c1) if c1 else None
    pw = ops.pack_weight(rnd(cout, cin, 3, 3) * 0.02, rnd(cout))
    g = ops.conv3x3_geom(n, hw, hw, stride=stride, up_to=(2 * hw, 2 * hw) if up else None)
    sweep(f"{name} n={n} {hw}x{hw} {cin}->{cout} s{stride}{' up' if up else ''}",
          lambda: ops.conv_gemm(x0, pw, g, x1=x1), 2.0 * g.rows * 9 * cin * cout, pw.n_pad)


def bench_tconv(name, hw, c):
    if not want(name):
        return
    x = rnd(N * hw * hw, c)
    pw = ops.pack_weight(rnd(c, c, 3, 1, 1) * 0.02, rnd(c))
    g = ops.tconv_geom(B, T, hw * hw)
    sweep(f"{name} {hw}x{hw} C={c}", lambda: ops.conv_gemm(x, pw, g), 2.0 * g.rows * 3 * c * c, pw.n_pad)


for c, hw in LEVELS[:3]:
    M = N * hw * hw
    bench_linear(f"linear qkv L{c}", M, c, 3 * c)
    bench_linear(f"linear proj L{c}", M, c, c)
    bench_linear(f"linear geglu L{c}", M, c, 8 * c, geglu=True)
    bench_linear(f"linear ff2 L{c}", M, 4 * c, c)
    bench_conv(f"conv3x3 L{c}", N, hw, c, c)
    bench_tconv(f"tconv L{c}", hw, c)
bench_conv("conv3x3 up L0 960->320", N, 64, 960, 320)
bench_conv("conv3x3 up L1 1920->640", N, 32, 1920, 640)
bench_conv("conv3x3 up L2 2560->1280", N, 16, 2560, 1280)
bench_conv("conv3x3 down L0 s2", N, 64, 320, 320, stride=2)
bench_conv("conv3x3 upsample L1->L0", N, 32, 640, 640, up=True)
bench_conv("conv3x3 L3", N, 8, 1280, 1280)

# ------------------------------------------------------------------ attention
for c, hw in LEVELS[:3]:
    name = f"attn spatial L{c}"
    if want(name):
        heads, L = c // 64, hw * hw
        qkv = rnd(N * L, 3 * c)
        fn = lambda: ops.attention(qkv, 0, qkv, c, qkv, 2 * c, heads, N, 1, L, L, (L, 0, 1), (L, 0, 1))
        report(f"{name} seq={L} heads={heads}", timeit(fn), flops=4.0 * N * heads * L * L * 64)
    name = f"attn temporal L{c}"
    if want(name):
        heads, P = c // 64, hw * hw
        qkv = rnd(N * P, 3 * c)
        st = (T * P, 1, P)
        fn = lambda: ops.attention(qkv, 0, qkv, c, qkv, 2 * c, heads, B, P, T, T, st, st)
        report(f"{name} seqs={B * P} heads={heads}", timeit(fn), flops=4.0 * B * P * heads * T * T * 64,
               bytes_=4.0 * N * P * c * 2)
    name = f"attn cross L{c}"
    if want(name):
        heads, L = c // 64, hw * hw
        q, kv = rnd(N * L, c), rnd(B * 77, 2 * c)
        fn = lambda: ops.attention(q, 0, kv, 0, kv, c, heads, N, 1, L, 77, (L, 0, 1), (77, 0, 1), kv_outer_div=T)
        report(f"{name} seq={L}x77 heads={heads}", timeit(fn), flops=4.0 * N * heads * L * 77 * 64,
               bytes_=2.0 * N * L * c * 2)

# ------------------------------------------------------------------ norms
# bytes: one read + one write of the tensor (what a GroupNorm has to move); each GroupNorm is timed as the single kernel the
# library picks (plan = groups per workgroup / pieces per thread / workgroups per image group / grid) and as the two-kernel pair
from animate_anything_amd import _lib as _lib_
for c, hw in LEVELS[:4] + [(640, 64), (960, 64), (1280, 32), (1920, 32), (2560, 16)]:
    tok = N * hw * hw
    for kind, ig, tpg in (("2d", N, hw * hw), ("3d", B, T * hw * hw)):
        name = f"groupnorm{kind}+silu C={c} {hw}x{hw}"
        if not want(name) or (kind == "3d" and c not in (320, 640, 1280)):
            continue
        x, gm, bt = rnd(tok, c), rnd(c), rnd(c)
        ops.GN_PLANS = []
        ops.groupnorm(x, gm, bt, ig, tpg, 32, 1e-5, True)
        plan, ops.GN_PLANS = ops.GN_PLANS[0], None
        report(f"{name} one kernel {plan}" if plan else f"{name} (two kernels: no one-kernel plan)",
               timeit(lambda: ops.groupnorm(x, gm, bt, ig, tpg, 32, 1e-5, True)), bytes_=2.0 * tok * c * 2)
        if plan:
            _lib_.get().aa_set_groupnorm_two_pass(1)
            report(f"{name} two kernels", timeit(lambda: ops.groupnorm(x, gm, bt, ig, tpg, 32, 1e-5, True)), bytes_=2.0 * tok * c * 2)
            _lib_.get().aa_set_groupnorm_two_pass(0)
    name = f"layernorm C={c} {hw}x{hw}"
    if want(name) and c in (320, 640, 1280) and (c, hw) in LEVELS:
        x, gm, bt = rnd(tok, c), rnd(c), rnd(c)
        report(name, timeit(lambda: ops.layernorm(x, gm, bt)), bytes_=2.0 * tok * c * 2)
