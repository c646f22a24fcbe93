#!/usr/bin/env python
"""The spatial self-attention calls of one 16 f x 512 x 512 step (34 images x 5 heads x 4096 tokens, 34 x 10 x 1024) under the kernel with two
32-query blocks per wave (round 6) and under the one-block kernel (AaAttention._pad bit 3): serialised single launches, median."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animate_anything_amd import ops  # noqa: E402

DT, dev = torch.float16, "cuda"


def timed(fn, reps=11):
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


for images, heads, L in [(34, 5, 4096), (34, 10, 1024), (34, 20, 256), (2, 5, 4096)]:
    C = heads * 64
    qkv = (torch.randn(images * L, 3 * C) * 1.0).to(DT).to(dev)
    st = (L, 0, 1)
    flops = 4.0 * images * heads * L * L * 64
    res = {}
    outs = {}
    for flags in (0, 8):
        ops.ATTN_FLAGS = flags
        fn = lambda: ops.attention(qkv, 0, qkv, C, qkv, 2 * C, heads, images, 1, L, L, st, st)
        outs[flags] = fn()
        res[flags] = timed(fn)
    err = (outs[0].float() - outs[8].float()).abs().max().item()
    print(f"images {images} heads {heads} tokens {L}: two blocks per wave {res[0]:.1f} us ({flops / res[0] * 1e-6:.0f} TF/s) | one block {res[8]:.1f} us "
          f"({flops / res[8] * 1e-6:.0f} TF/s) | max |difference| {err:.5f}")
ops.ATTN_FLAGS = 0
