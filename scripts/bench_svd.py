#!/usr/bin/env python
"""Denoising steps/sec of the Stable-Video-Diffusion path at BASELINE.json configs[3] (14 frames x 576x1024, bs=1): a
side measurement next to bench.py (whose metric stays the 16x512x512 UNet3D step).

One step = one iteration of the reference's loop (models/pipeline.py:413-451): input assembly + the
UNetSpatioTemporalConditionModel forward on the CFG-doubled batch (28 images x 72x128 latents) + per-frame guidance +
the Euler update.  Seeded random weights of the stable-video-diffusion-img2vid architecture with the reference's 9 input
channels (1.5 G parameters), synthetic latents / image embedding, inputs resident in HBM.

    python scripts/bench_svd.py [--steps 10 --warmup 3] [--tile-cache FILE] [--gemm-breakdown FILE]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MFMA_PEAK_TFLOPS = 2500.0


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    p.add_argument("--frames", type=int, default=14)
    p.add_argument("--height", type=int, default=576)
    p.add_argument("--width", type=int, default=1024)
    p.add_argument("--text-len", type=int, default=1, help="context tokens per clip (1 = CLIP image embedding, 77 = text)")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--tile-cache", default="")
    p.add_argument("--gemm-breakdown", default="")
    a = p.parse_args()
    assert torch.cuda.is_available(), "needs an MI355X (no CPU fallback)"
    from animate_anything_amd import ops
    from animate_anything_amd.schedulers import EulerDiscreteScheduler
    from animate_anything_amd.svd_pipeline import StableVideoDiffusionPipeline
    from animate_anything_amd.svd_unet import UNetSpatioTemporalConditionModel
    dtype = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    with torch.device(dev):
        unet = UNetSpatioTemporalConditionModel(in_channels=9, num_frames=a.frames).to(dtype).eval()
    with torch.no_grad():
        for n_, p_ in unet.named_parameters():
            if n_.endswith("mix_factor"):
                p_.fill_(0.3)
    if a.tile_cache and os.path.exists(a.tile_cache):
        ops.load_tile_cache(a.tile_cache)
    if not a.no_graph:
        unet.enable_graph()
    pipe = StableVideoDiffusionPipeline(None, None, unet, EulerDiscreteScheduler())
    g = torch.Generator(device=dev).manual_seed(1234)
    h, w, f = a.height // 8, a.width // 8, a.frames
    latents = torch.randn(1, f, 4, h, w, generator=g, device=dev)
    cond = torch.randn(1, 1, 4, h, w, generator=g, device=dev).repeat(1, f, 1, 1, 1).to(dtype)
    cond = torch.cat([torch.zeros_like(cond), cond])
    emb = torch.randn(1, a.text_len, 1024, generator=g, device=dev).to(dtype)
    emb = torch.cat([torch.zeros_like(emb), emb])
    mask = torch.zeros(2, f, 1, h, w, device=dev, dtype=dtype)
    mask[..., h // 4: h - h // 4, w // 4: w - w // 4] = 1
    ids = torch.tensor([[6.0, 127.0, 0.02]], device=dev).repeat(2, 1)
    guidance = torch.linspace(1.0, 3.0, f)
    total = a.warmup + a.steps
    pipe.scheduler.set_timesteps(max(total, 2))
    ts = pipe.scheduler.timesteps[:total]

    def run(tsteps, x):
        return pipe.denoise(x, emb, ids, cond, mask, guidance, tsteps)

    with torch.no_grad():
        x = run(ts[: a.warmup], latents * pipe.scheduler.init_noise_sigma) if a.warmup else latents
        if a.tile_cache:
            ops.save_tile_cache(a.tile_cache)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = run(ts[a.warmup:], x)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert torch.isfinite(x).all(), "non-finite latents"

    # FLOP census of one step from the launches themselves: contractions from their descriptors, attention analytically
    unet.enable_graph(False)
    ops.TRACE = []
    att = []
    real_attention = ops.attention

    def counting_attention(q, q0, k, k0, v, v0, heads, n_outer, n_inner, q_len, kv_len, *r, **kw):
        att.append(4.0 * n_outer * n_inner * heads * q_len * kv_len * kw.get("head_dim", 64))
        return real_attention(q, q0, k, k0, v, v0, heads, n_outer, n_inner, q_len, kv_len, *r, **kw)

    ops.attention = counting_attention
    with torch.no_grad():
        run(ts[:1], latents)
    ops.attention = real_attention
    trace, ops.TRACE = ops.TRACE, None
    torch.cuda.synchronize()
    gemm_flop = sum(2.0 * d.n_img * d.h_out * d.w_out * d.n_out * d.kh * d.kw * (d.c0 + d.c1) for d, _ in trace)
    flop_step = gemm_flop + sum(att)
    import ctypes as C
    from animate_anything_amd import _lib
    lib = _lib.get()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for d, _ in trace:
        lib.aa_conv_gemm(C.byref(d), stream)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(2):
        for d, _ in trace:
            lib.aa_conv_gemm(C.byref(d), stream)
    e1.record()
    torch.cuda.synchronize()
    gemm_ms = e0.elapsed_time(e1) / 2
    if a.gemm_breakdown:
        from collections import defaultdict
        groups = defaultdict(lambda: [0, 0.0, 0.0, -1])
        for d, _ in trace:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(3):
                lib.aa_conv_gemm(C.byref(d), stream)
            ev1.record()
            ev1.synchronize()
            key = (d.n_img * d.h_out * d.w_out, d.kh * d.kw * (d.c0 + d.c1), d.n_out, f"{d.kh}x{d.kw}s{d.stride}",
                   "geglu" if d.geglu else "", "up" if d.h_virt != d.h_in else "")
            g_ = groups[key]
            g_[0] += 1
            g_[1] += ev0.elapsed_time(ev1) / 3
            g_[2] += 2.0 * key[0] * key[1] * key[2]
            g_[3] = f"{d.tile}/{d.k_splits}" if d.k_splits else f"{d.tile}"
        with open(a.gemm_breakdown, "w") as fo:
            fo.write("M K N kind flags | launches total_ms TFLOP/s tile share\n")
            tot = sum(v[1] for v in groups.values())
            for key, v in sorted(groups.items(), key=lambda kv: -kv[1][1]):
                fo.write(f"{key[0]:7d} {key[1]:6d} {key[2]:6d} {key[3]:6s} {key[4]:5s}{key[5]:3s} | {v[0]:3d} {v[1]:8.3f} "
                         f"{v[2] / (v[1] * 1e-3) / 1e12:7.1f} {v[3]:>4s} {v[1] / tot * 100:5.1f}%\n")
    ms = dt / a.steps * 1e3
    out = {"metric": f"SVD denoising steps/sec @{f}fx{a.height}x{a.width} bs=1", "value": round(a.steps / dt, 4), "unit": "steps/s",
           "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
           "dtype": a.dtype, "data": "synthetic",
           "config": {"workload": f"stable-video-diffusion-img2vid UNetSpatioTemporalConditionModel, 9 input channels "
                                  f"({sum(p_.numel() for p_ in unet.parameters()) / 1e6:.0f}M params, seeded random init), {f} frames x "
                                  f"{a.height}x{a.width}, CFG batch 2, context {a.text_len} token(s), Euler step, "
                                  f"hipGraph={'off' if a.no_graph else 'on'}"},
           "flop_per_step": flop_step, "attention_flop_per_step": sum(att), "tflops": round(flop_step / (ms * 1e-3) / 1e12, 1),
           "whole_step_frac_of_peak": round(flop_step / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
           "contraction_kernel": {"launches_per_step": len(trace), "ms_per_step": round(gemm_ms, 3),
                                  "tflops": round(gemm_flop / (gemm_ms * 1e-3) / 1e12, 1)}}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
