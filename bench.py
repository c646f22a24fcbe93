#!/usr/bin/env python
"""bench.py -- UNet3D denoising steps/sec @ 16 frames x 512x512, bs=1 (BASELINE.json metric).

One "step" = one iteration of the reference's denoising loop (models/pipeline.py:163-198): the
UNet3DConditionModel forward on the CFG-doubled batch [2,4,16,64,64] (+ condition frame => 17 frames
inside), classifier-free guidance and the DPM-Solver++ update.  Synthetic seeded weights of the
v1.02 architecture (1413 M parameters) and synthetic latents/text; inputs are
resident in HBM before the timed region.  With N GPUs every rank denoises its own clip (weak scaling,
no data-path collective) and the final latents are all-gathered over RCCL once at the end.

Prints ONE JSON line (rank 0).  `roofline` re-times the dominant kernel (the implicit-GEMM
contraction, conv_gemm_kernel) in isolation with events on its own stream; `cpu_baseline` times the
CPU oracle (oracle/, a restatement of the reference: "port") on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_STEP = 44.262e12        # BASELINE.md section 2 census (2*MAC of every conv/linear/attention matmul)
MFMA_PEAK_TFLOPS = 2500.0        # gfx950 dense bf16/fp16 (MI355X_MICROARCH.md)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    p.add_argument("--frames", type=int, default=16)
    p.add_argument("--size", type=int, default=512)
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--tile-cache", default="", help="json file with autotuned tile choices: loaded if present, written after warm-up")
    p.add_argument("--gemm-breakdown", default="", help="write a per-shape table of the contraction launches of one step")
    return p.parse_args()


def build_unet(dtype, device):
    from animate_anything_amd.unet3d import UNet3DConditionModel
    torch.manual_seed(0)
    with torch.device(device):
        net = UNet3DConditionModel(motion_mask=True, motion_strength=True)
    # the architecture zero-initialises TemporalConvLayer.conv4 / motion_embedding[-1]: re-draw them so
    # no path is vacuous
    with torch.no_grad():
        for p_ in net.parameters():
            if p_.abs().max() == 0:
                p_.normal_(0.0, 0.02)
    return net.to(dtype).eval()


def synthetic_inputs(frames, lat, dtype, device, seed=1234):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    mask = torch.zeros(1, 1, 1, lat, lat)
    mask[..., lat // 4: lat - lat // 4, lat // 4: lat - lat // 4] = 1
    d = dict(latents=r(1, 4, frames, lat, lat), cond=r(1, 4, 1, lat, lat), mask=mask,
             text=r(1, 77, 1024), neg=r(1, 77, 1024))
    return {k: v.to(device=device, dtype=dtype if k != "latents" else torch.float32) for k, v in d.items()}


def cpu_baseline(frames_sample, lat_full, lat=32):
    """Oracle UNet forward (fp32 CPU, `cores` threads) on a bounded sample of the workload: the CFG batch
    of 2, `frames_sample`+1 of the 17 frames, at `lat`x`lat` instead of 64x64 latents; steps/s scaled by
    the token ratio (all per-token costs are linear except spatial attention, which this under-counts)."""
    import oracle
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    net = oracle.UNet3DConditionModel(motion_mask=True, motion_strength=True).eval()
    g = torch.Generator().manual_seed(1234)
    r = lambda *s: torch.randn(*s, generator=g)
    x, c, txt = r(2, 4, frames_sample, lat, lat), r(2, 4, 1, lat, lat), r(2, 77, 1024)
    m = torch.zeros(1, 1, 1, lat, lat)
    t0 = time.perf_counter()
    with torch.no_grad():
        net(x, 500, txt, c, m, motion=torch.tensor([3.0]))
    dt = time.perf_counter() - t0
    scale = 17.0 / (frames_sample + 1) * (lat_full / lat) ** 2
    return dict(value=1.0 / (dt * scale), unit="steps/s", cores=cores, kind="port",
                sample=f"oracle UNet3D forward fp32 on {cores} threads, CFG batch 2, {frames_sample}+1 of 16+1 frames at "
                       f"{lat}x{lat} (of {lat_full}x{lat_full}) latents, {dt:.1f} s measured, scaled x{scale:.1f} by token count")


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)
    dtype = torch.float16 if a.dtype == "fp16" else torch.bfloat16

    from animate_anything_amd import ops
    from animate_anything_amd.pipeline import LatentToVideoPipeline
    from animate_anything_amd.schedulers import DPMSolverMultistepScheduler

    lat = a.size // 8
    unet = build_unet(dtype, device)
    pipe = LatentToVideoPipeline(vae=None, unet=unet, scheduler=DPMSolverMultistepScheduler())
    inp = synthetic_inputs(a.frames, lat, dtype, device, seed=1234 + rank)      # one clip per rank
    embeds = torch.cat([inp["neg"], inp["text"]])
    total = a.warmup + a.steps
    pipe.scheduler.set_timesteps(max(total, 2))
    ts = [int(t) for t in pipe.scheduler.timesteps][:total]
    if not a.no_graph:
        unet.enable_graph()
    if a.tile_cache and os.path.exists(a.tile_cache):
        ops.load_tile_cache(a.tile_cache)

    def run(tsteps, x):
        return pipe.denoise(x, embeds, inp["cond"], inp["mask"], [3.0], tsteps, 9.0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        x = run(ts[: a.warmup], inp["latents"]) if a.warmup else inp["latents"]
        if a.tile_cache and rank == 0:
            ops.save_tile_cache(a.tile_cache)
        barrier()
        t0 = time.perf_counter()
        x = run(ts[a.warmup:], x)
        if world > 1:
            gathered = torch.empty((world,) + tuple(x.shape[1:]), dtype=dtype, device=device)
            dist.all_gather_into_tensor(gathered, x.to(dtype).contiguous())
        barrier()
        dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    assert torch.isfinite(x).all(), "non-finite latents"

    ms_step = dt / a.steps * 1e3
    value = world * a.steps / dt
    out = {
        "metric": "UNet3D denoising steps/sec @16fx512x512 bs=1", "value": round(value, 4), "unit": "steps/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"animate_anything_512_v1.02 UNet3D (1413M params, seeded random init), "
                               f"{a.frames} frames x {a.size}x{a.size}, pipeline bs=1 per GPU (CFG batch 2, 17 frames "
                               f"inside), DPM-Solver++ step, hipGraph={'off' if a.no_graph else 'on'}",
                   "clips_per_gpu": 1, "parallelism": f"clip-sharded x{world}"},
        "tflops_per_gpu": round(FLOP_PER_STEP * (a.steps / dt) / 1e12 * (a.frames + 1) / 17 * (lat / 64) ** 2, 2),
    }

    if rank == 0 and not a.no_roofline:
        # dominant kernel in isolation: record one eager step's contraction launches, replay them
        # back-to-back on the current stream between two events (same stream the kernels run on).
        unet.enable_graph(False)
        ops.TRACE = []
        with torch.no_grad():
            run(ts[:1], inp["latents"])
        trace, ops.TRACE = ops.TRACE, None
        torch.cuda.synchronize()
        from animate_anything_amd import _lib
        import ctypes as C
        lib = _lib.get()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        flops = sum(2.0 * d.n_img * d.h_out * d.w_out * d.n_out * d.kh * d.kw * (d.c0 + d.c1) for d, _ in trace)
        reps = 3
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for d, _ in trace:
            lib.aa_conv_gemm(C.byref(d), stream)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            for d, _ in trace:
                lib.aa_conv_gemm(C.byref(d), stream)
        e1.record()
        torch.cuda.synchronize()
        gemm_ms = e0.elapsed_time(e1) / reps
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
            with open(a.gemm_breakdown, "w") as f:
                f.write("M K N kind flags | launches total_ms TFLOP/s tile share\n")
                tot = sum(v[1] for v in groups.values())
                for key, v in sorted(groups.items(), key=lambda kv: -kv[1][1]):
                    f.write(f"{key[0]:7d} {key[1]:6d} {key[2]:6d} {key[3]:6s} {key[4]:5s}{key[5]:3s} | {v[0]:3d} {v[1]:8.3f} "
                            f"{v[2] / (v[1] * 1e-3) / 1e12:7.1f} {v[3]:>4s} {v[1] / tot * 100:5.1f}%\n")
        ach = flops / (gemm_ms * 1e-3) / 1e12
        # HBM bytes per launch of the dominant kernel: PMC numbers (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE,
        # separate rocprofv3 passes, scripts/pmc_traffic.sh) committed under profiles/ - bench.py cannot run the
        # profiler on itself, so `traffic` is the last committed measurement of the same command (null if absent)
        traffic, tpath = None, os.path.join(ROOT, "profiles", "r01_traffic_pmc.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("conv_gemm_dma_kernel", {}).get("hbm_bytes_per_launch")
        out["roofline"] = {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                           "traffic_unit": "HBM bytes per launch (PMC, profiles/r01_traffic_pmc.json)",
                           "kernel": "aa::conv_gemm_kernel (implicit-GEMM conv/linear, all instances)",
                           "launches_per_step": len(trace), "avg_launch_us": round(gemm_ms * 1e3 / len(trace), 2),
                           "flop_per_step_in_kernel": flops, "kernel_ms_per_step": round(gemm_ms, 3),
                           "whole_step_frac_of_peak": round(FLOP_PER_STEP * (a.steps / dt) / 1e12 / MFMA_PEAK_TFLOPS, 4)}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(1, lat)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
