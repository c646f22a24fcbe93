#!/usr/bin/env python
"""How much does the sparsely filled last residency round of the 4096-key self-attention cost?  The step's call has 32 x 5 x 34 = 5440 workgroups
on 768 resident slots (3 per CU) = 7.083 rounds.  Time the kernel at (heads x sequences) products around the round boundaries."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animate_anything_amd import ops
L, dt = 4096, torch.float16
def t(heads, n, reps=10):
    c = heads * 64
    qkv = torch.randn(n * L, 3 * c, device="cuda").to(dt)
    fn = lambda: ops.attention(qkv, 0, qkv, c, qkv, 2 * c, heads, n, 1, L, L, (L, 0, 1), (L, 0, 1))
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for heads, n in ((8, 21), (5, 34), (8, 24), (5, 38), (6, 32), (8, 18), (5, 29), (4, 36), (5, 24)):
    units = 32 * heads * n
    us = t(heads, n)
    print(f"heads {heads} sequences {n:3d}: {units:5d} workgroups = {units / 768:6.3f} rounds of 768: {us:8.1f} us = {us / units * 768:7.1f} us per 768 workgroups, {us / -(-units // 768):7.1f} us per started round")
