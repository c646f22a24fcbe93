#!/usr/bin/env python
"""aa_linear_rows ([LayerNorm](x) W^T + b (+ residual), 320-channel rows in registers) against the contraction of the tile family it replaces
(aa_conv_gemm with the LayerNorm folded from producer-written row coefficients), at the shapes of one 16 f x 512 x 512 step: serialised
single launches (event pair around each, median of medians, forms interleaved), random data."""
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


def case(rows, n_out, ln, with_res, C=320):
    g = torch.Generator(device="cpu").manual_seed(0)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DT).to(dev)
    xin = r(rows, C)
    w, b = r(n_out, C, sc=C ** -0.5), r(n_out, sc=0.1)
    wo, bo = r(C, C, sc=C ** -0.5), r(C, sc=0.1)
    res = r(rows, n_out) if with_res else None
    gamma, beta = (1.0 + 0.1 * r(C).float()).to(DT), r(C, sc=0.1)
    lnp = (gamma, beta, 1e-5) if ln else None
    pk = ops.pack_linear_rows(w, b, ln=lnp)
    x, stats = ops.conv_gemm(xin, ops.pack_weight(wo, bo), ops.linear_geom(rows), residual=xin, row_stats=True, coef_eps=1e-5)
    pw = ops.pack_weight(w, b, ln=lnp)
    rows_fn = lambda: ops.linear_rows(x, pk, res)
    tile_fn = lambda: ops.conv_gemm(x, pw, ops.linear_geom(rows), residual=res, **({"ln_stats": stats} if ln else {}))
    a, t = rows_fn(), tile_fn()
    err = (a.float() - t.float()).abs().max().item()
    for _ in range(20):
        rows_fn(); tile_fn()
    ta, tb = [], []
    for _ in range(3):
        ta.append(timed(rows_fn, 9)); tb.append(timed(tile_fn, 9))
    t_a, t_b = sorted(ta)[1], sorted(tb)[1]
    mb = rows * (C + n_out * (2 if with_res else 1)) * 2e-6
    flops = 2.0 * rows * C * n_out
    print(f"rows={rows} 320->{n_out} ln={int(ln)} res={int(with_res)}: linear_rows {t_a:.1f} us ({flops / t_a * 1e-6:.0f} TF/s, {mb / t_a:.2f} TB/s) | "
          f"tile family {t_b:.1f} us ({flops / t_b * 1e-6:.0f} TF/s) | max |diff| {err:.4f} (|out| max {t.float().abs().max().item():.2f})", flush=True)


if __name__ == "__main__":
    for rows in (139264, 69632):
        case(rows, 320, False, True)      # attention out-projection + residual
        case(rows, 320, True, False)      # to_q of the text cross-attention behind norm2
        case(rows, 320, False, False)     # proj_in
        case(rows, 960, True, False)      # Q|K|V of the spatial self-attention behind norm1
