#!/usr/bin/env python
"""Yardstick only (never used by the product): what the vendor GEMM library (torch.matmul -> hipBLASLt / rocBLAS) reaches
on the plain-GEMM equivalents of this step's contractions (im2col'ed K for the convs, no fused epilogue)."""
import torch

dev = "cuda"
shapes = [(139264, 320, 320), (139264, 320, 960), (139264, 320, 2560), (139264, 1280, 320), (139264, 2880, 320),
          (34816, 640, 640), (34816, 640, 5120), (34816, 5760, 640), (8704, 1280, 1280), (8704, 11520, 1280), (2176, 11520, 1280)]
for M, K, N in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16)
    for _ in range(3):
        torch.matmul(a, w.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        torch.matmul(a, w.t())
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"M={M:7d} K={K:6d} N={N:6d}  {us:9.1f} us  {2.0 * M * K * N / us / 1e6:8.1f} TFLOP/s", flush=True)
