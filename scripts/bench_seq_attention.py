#!/usr/bin/env python
"""aa_seq_self_attention (LayerNorm + Q|K|V + 17-frame attention in one kernel) against the three-launch form it replaces
(Q|K|V contraction with the LayerNorm folded, aa_attention on strided rows) at the temporal-transformer shapes of one
16 f x 512 x 512 step: serialised single launches (event pair around each, median), random data."""
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


def case(C, clips, frames, hw):
    rows = clips * frames * hw
    g = torch.Generator(device="cpu").manual_seed(0)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DT).to(dev)
    x = r(rows, C)
    wq, wk, wv = (r(C, C, sc=C ** -0.5) for _ in range(3))
    gamma, beta = (1.0 + 0.1 * r(C).float()).to(DT), r(C, sc=0.1)
    wo, bo = r(C, C, sc=C ** -0.5), r(C, sc=0.1)
    st = (frames * hw, 1, hw)
    w_seq = ops.pack_seq_qkv(wq, wk, wv, ln=(gamma, beta, 1e-5))
    fused = lambda: ops.seq_self_attention(x, w_seq, clips, hw, frames, st)
    # the form it replaces: LayerNorm folded into the Q|K|V contraction (row statistics from the producer), attention on strided rows
    pw = ops.pack_weight(torch.cat([wq, wk, wv]), None, ln=(gamma, beta, 1e-5))
    xin = x
    x, stats = ops.conv_gemm(xin, ops.pack_weight(wo, bo), ops.linear_geom(rows), residual=xin, row_stats=True, coef_eps=1e-5)     # (x = a tensor WITH the row statistics of its producer)
    qkv_call = lambda: ops.conv_gemm(x, pw, ops.linear_geom(rows), ln_stats=stats)
    qkv = qkv_call()
    attn_call = lambda: ops.attention(qkv, 0, qkv, C, qkv, 2 * C, C // 64, clips, hw, frames, frames, st, st)
    a = attn_call()
    f = fused()
    err = (f.float() - a.float()).abs().max().item()
    pwo = ops.pack_weight(wo, bo)
    out_call = lambda: ops.conv_gemm(a, pwo, ops.linear_geom(rows), residual=x)
    t_f, t_q, t_a, t_o = timed(fused), timed(qkv_call), timed(attn_call), timed(out_call)
    # with the linear layer in front inside the kernel (the previous attention layer's to_out + residual, or proj_in)
    pre = ops.pack_seq_pre(wo, bo)
    t_p = timed(lambda: ops.seq_self_attention(a, w_seq, clips, hw, frames, st, pre=pre, residual=x))
    flops = 2.0 * rows * C * 3 * C + 4.0 * rows * frames * C
    print(f"C={C} rows={rows} (clips {clips}, frames {frames}, hw {hw}): fused {t_f:.1f} us ({flops / t_f * 1e-6:.0f} TF/s, {2 * rows * C * 2 / t_f * 1e-3:.0f} GB/s x+o)"
          f" | Q|K|V {t_q:.1f} + attention {t_a:.1f} = {t_q + t_a:.1f} us | to_out {t_o:.1f} us | to_out + residual inside the fused kernel {t_p:.1f} us"
          f" (apart {t_o + t_f:.1f}) | max |fused - three-launch| {err:.4f}")


def ablations(C, clips, frames, hw):
    """Where a call's time goes: AaSeqSelfAttn.flags switch parts of the kernel off (results are garbage)."""
    rows = clips * frames * hw
    g = torch.Generator(device="cpu").manual_seed(0)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DT).to(dev)
    x = r(rows, C)
    gamma, beta = (1.0 + 0.1 * r(C).float()).to(DT), r(C, sc=0.1)
    w_seq = ops.pack_seq_qkv(*(r(C, C, sc=C ** -0.5) for _ in range(3)), ln=(gamma, beta, 1e-5))
    st = (frames * hw, 1, hw)
    for flags, what in [(0, "whole kernel"), (1, "no attention phase"), (2, "no weight DMA behind the first two stages"), (4, "no projection MFMAs"),
                        (8, "no x fetch / LayerNorm"), (16, "no output stores"), (1 | 4, "no attention, no projection MFMAs"), (1 | 2 | 4, "+ no weight DMA"),
                        (1 | 2 | 4 | 8 | 16, "barriers and loops only"), (2 | 8 | 16, "no global traffic at all"), (8 | 16, "no x fetch, no stores"),
                        (32, "no row normalisation (x fetched)")]:
        ops.SEQ_ATTN_DEBUG = flags
        t = timed(lambda: ops.seq_self_attention(x, w_seq, clips, hw, frames, st))
        print(f"  C={C} rows={rows} flags {flags:2d} ({what}): {t:.1f} us")
    ops.SEQ_ATTN_DEBUG = 0


if __name__ == "__main__":
    if "--ablate" in sys.argv:
        ablations(320, 2, 17, 4096)
        sys.exit(0)
    case(320, 2, 17, 4096)
    case(640, 2, 17, 1024)
    case(512, 2, 17, 4096)
    case(320, 1, 17, 4096)
    case(320, 2, 9, 1024)
