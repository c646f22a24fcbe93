#!/usr/bin/env python
"""Phase stamps (scripts/phase_probe_x.py) of the projections that consume a folded LayerNorm, next to the same projection without
the fold: where in a workgroup's life does the fold cost its 10-19 %?  (r04; AaConvGemm.debug bit 8)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import phase_probe_x as P  # noqa: E402  (runs its own table first; PROBE_ONLY=none skips it)
from animate_anything_amd import ops  # noqa: E402

M = 34 * 64 * 64
x = P.rnd(M, 320) + 1.0
xf = x.float()
st = ops.RowStats(torch.stack([xf.sum(1), (xf * xf).sum(1)], dim=1).reshape(M, 1, 2).contiguous(), M, 1)
g, b = P.rnd(320) * 0.3 + 1.0, P.rnd(320) * 0.2
wg, bg = P.rnd(2560, 320) * 0.05, P.rnd(2560)
pw_plain = ops.pack_weight(wg, bg, geglu=True)
pw_fold = ops.pack_weight(wg, bg, geglu=True, ln=(g, b, 1e-5))
os.environ["PROBE_ONLY"] = ""
P.ONLY = ""
P.probe("geglu K=320 N=2560 plain", lambda: ops.conv_gemm(x, pw_plain, ops.linear_geom(M)), [36, 38])
P.probe("geglu K=320 N=2560 LN folded", lambda: ops.conv_gemm(x, pw_fold, ops.linear_geom(M), ln_stats=st), [36, 38])
wq = P.rnd(960, 320) * 0.05
pq_plain, pq_fold = ops.pack_weight(wq), ops.pack_weight(wq, None, ln=(g, b, 1e-5))
P.probe("qkv K=320 N=960 plain", lambda: ops.conv_gemm(x, pq_plain, ops.linear_geom(M)), [39, 49])
P.probe("qkv K=320 N=960 LN folded", lambda: ops.conv_gemm(x, pq_fold, ops.linear_geom(M), ln_stats=st), [39, 49])
