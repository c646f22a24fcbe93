#!/usr/bin/env python
"""aa_ff_fused (LayerNorm + GEGLU.proj + value * gelu(gate) + ff-out + residual + proj_out + residual in one kernel) against the two
contractions it replaces (GEGLU contraction with the LayerNorm folded, merged ff-out / proj_out two-source contraction) at the 64 x 64
level of one 16 f x 512 x 512 step: serialised single launches (event pair around each, median), random data."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animate_anything_amd import ops  # noqa: E402

DT = torch.float16
dev = "cuda"


def timed(fn, reps=15):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def case(rows, C=320):
    g = torch.Generator(device="cpu").manual_seed(0)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DT).to(dev)
    xin, outer = r(rows, C), r(rows, C)
    w1, b1 = r(8 * C, C, sc=C ** -0.5), r(8 * C, sc=0.1)
    w2, b2 = r(C, 4 * C, sc=(4 * C) ** -0.5), r(C, sc=0.1)
    wp, bp = r(C, C, sc=C ** -0.5), r(C, sc=0.1)
    wo, bo = r(C, C, sc=C ** -0.5), r(C, sc=0.1)
    gamma, beta = (1.0 + 0.1 * r(C).float()).to(DT), r(C, sc=0.1)
    pk = ops.pack_ff_fused(w1, b1, w2, b2, wp, bp, ln=(gamma, beta, 1e-5))
    # the form it replaces: x with the row statistics of its producer, GEGLU with the LayerNorm folded, [h | x] against the merged tail
    x, stats = ops.conv_gemm(xin, ops.pack_weight(wo, bo), ops.linear_geom(rows), residual=xin, row_stats=True, coef_eps=1e-5)
    fused = lambda: ops.ff_fused(x, pk, outer)
    pw1 = ops.pack_weight(w1, b1, geglu=True, ln=(gamma, beta, 1e-5))
    wpf = wp.float()
    wm = torch.cat([wpf @ w2.float(), wpf], dim=1).to(DT)
    bm = (bp.float() + wpf @ b2.float()).to(DT)
    pwm = ops.pack_weight(wm, bm)
    geglu = lambda: ops.conv_gemm(x, pw1, ops.linear_geom(rows), ln_stats=stats)
    h = geglu()
    tail = lambda: ops.conv_gemm(h, pwm, ops.linear_geom(rows), x1=x, residual=outer)
    two = tail()
    f = fused()
    err = (f.float() - two.float()).abs().max().item()
    for _ in range(20):
        fused(); geglu(); tail()
    tf, tg, tt = [], [], []
    for _ in range(3):                                  # (the clock under load drifts over the first seconds: interleave the forms)
        tf.append(timed(fused, 9)); tg.append(timed(geglu, 9)); tt.append(timed(tail, 9))
    t_f, t_g, t_t = sorted(tf)[1], sorted(tg)[1], sorted(tt)[1]
    flops = 2.0 * rows * C * (8 * C + 4 * C + C)
    print(f"C={C} rows={rows}: fused {t_f:.1f} us ({flops / t_f * 1e-6:.0f} TF/s) | GEGLU {t_g:.1f} + merged tail {t_t:.1f} = {t_g + t_t:.1f} us "
          f"({flops / (t_g + t_t) * 1e-6:.0f} TF/s) | max |fused - two-launch| {err:.4f} (|out| max {two.float().abs().max().item():.2f})")


def ablations(rows, C=320):
    g = torch.Generator(device="cpu").manual_seed(0)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DT).to(dev)
    x, outer = r(rows, C), r(rows, C)
    pk = ops.pack_ff_fused(r(8 * C, C, sc=C ** -0.5), r(8 * C, sc=0.1), r(C, 4 * C, sc=(4 * C) ** -0.5), r(C, sc=0.1), r(C, C, sc=C ** -0.5), r(C, sc=0.1),
                           ln=((1.0 + 0.1 * r(C).float()).to(DT), r(C, sc=0.1), 1e-5))
    A = lambda bits: bits << 8
    variants = [(0, "whole kernel"), (2, "weight DMA issued, fetches nothing"), (A(4), "no weight DMA issued"), (A(1), "no GELU arithmetic"),
                (A(32), "no bias k-slice"), (A(33), "no GELU, no bias k-slice"), (A(8), "no fragment reads"), (A(16), "no MFMAs"),
                (A(24), "no fragment reads, no MFMAs"), (A(61), "barriers, loops and the x / out traffic only"),
                (A(256), "variant: fine side work (a third of a GELU pair behind every MFMA)"), (A(64), "variant: GELU two pairs per slot"),
                (A(128), "variant: no rotation of the DMA piece order"), (A(512), "variant: x / outer / out non-temporal")]
    # three passes over all variants (the chip's clock under load drifts over the first seconds: a single pass favours whatever runs last)
    for _ in range(30):
        ops.ff_fused(x, pk, outer)
    res = {f: [] for f, _ in variants}
    for rep in range(3):
        for flags, what in variants:
            ops.FF_FUSED_DEBUG = flags
            res[flags].append(timed(lambda: ops.ff_fused(x, pk, outer), reps=9))
    for flags, what in variants:
        print(f"  rows={rows} flags {flags:6d} ({what}): " + " / ".join(f"{t:.1f}" for t in res[flags]) + " us")
    ops.FF_FUSED_DEBUG = 0


if __name__ == "__main__":
    if "--ablate" in sys.argv:
        ablations(139264)
        sys.exit(0)
    case(139264)
    case(69632)
    case(34816)
