#!/usr/bin/env python
"""bench.py -- UNet3D denoising steps/sec @ 16 frames x 512x512, bs=1 (BASELINE.json metric).

One "step" = one iteration of the reference's denoising loop (models/pipeline.py:163-198): the
UNet3DConditionModel forward on the CFG-doubled batch [2,4,16,64,64] (+ condition frame => 17 frames
inside), classifier-free guidance and the DPM-Solver++ update.  Synthetic seeded weights of the
v1.02 architecture (1413 M parameters) and synthetic latents/text; inputs are
resident in HBM before the timed region.  With N GPUs every rank denoises its own clip (weak scaling,
no data-path collective) and the final latents are all-gathered over RCCL once at the end.

`python bench.py --gpus N` with N > 1 and no torchrun environment re-executes itself under
`python -m torch.distributed.run --nproc-per-node N` (one rank per GPU, RCCL); under torchrun `--gpus` must equal
WORLD_SIZE.  Clip ownership / seeds / the final gather are animate_anything_amd.distributed.

`--workload svd` / `--workload rgba` print the same JSON schema (with `roofline` and `cpu_baseline`) for BASELINE.json
configs[3] (Stable-Video-Diffusion UNet, 14 frames x 576x1024) and configs[4] (the layerdiffuse RGBA path: the same UNet3D at
16 frames x 384x384 plus its transparent-VAE add-ons, timed once per clip); the driver's default command stays configs[1].

Prints ONE JSON line (rank 0).  `roofline` re-times the dominant kernel (the implicit-GEMM
contraction: aa::conv_gemm_dma_kernel / conv3x3_slab_kernel / conv_gemm_x_kernel) in isolation with events on its own stream; `cpu_baseline` times the
CPU oracle (oracle/, a restatement of the reference: "port") on BASELINE.json configs[0] (8 frames x 256x256, the
reference's own CPU-runnable case) - a full forward, median of 3 - and quotes the one full 16x512x512 forward measured
offline on the same class of host (profiles/r02_cpu_baseline.json).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_STEP = 44.262e12        # BASELINE.md section 2 census (2*MAC of every conv/linear/attention matmul), 16f x 512x512, CFG batch 2
FLOP_TABLE = {(16, 64): 44.262e12, (16, 48): 23.932e12, (8, 32): 5.495e12}     # (frames, latent size) -> FLOP per step
# The census counts the reference's arithmetic: Upsample2D = nearest x2 + a 3x3 convolution on the upsampled grid (9 taps per output pixel).  The
# product runs it as four 2x2 convolutions on the input grid (4 taps per output pixel: ops.pack_upsample2x_weights), so 5/9 of those three
# convolutions is never executed: 2 * 34 images * (256 * 1280^2 + 1024 * 1280^2 + 4096 * 640^2) * 9 * 5/9 at 16 f x 512 x 512.
UPSAMPLE_FLOP_NOT_EXECUTED = 2.0 * 34 * (256 * 1280 ** 2 + 1024 * 1280 ** 2 + 4096 * 640 ** 2) * 5
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
    p.add_argument("--no-other-form", action="store_true", help="skip the second timing (other guidance form) and the 2-step cross-check: profiling runs")
    p.add_argument("--cfg-shared-prefix", action="store_true",
                   help="time the product default (text-independent UNet prefix computed once per guidance pair) as the headline "
                        "instead of the strict form that recomputes it for both halves like the reference")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-vae", action="store_true", help="skip the per-clip AutoencoderKL encode / decode timing (the `vae` object)")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--no-tile-cache", action="store_true", help="ignore the committed tile choices (animate_anything_amd/tile_cache_gfx950.json): autotune everything")
    p.add_argument("--tile-cache", default="", help="json file with autotuned tile choices: loaded if present, written after warm-up")
    p.add_argument("--gemm-breakdown", default="", help="write a per-shape table of the contraction launches of one step")
    p.add_argument("--launch-list", default="", help="run one more eager step and write its contraction calls IN LAUNCH ORDER (shape, kernel launches, "
                                                     "algorithmic bytes) as json: what scripts/pmc_traffic_report.py joins the per-dispatch PMC rows with")
    p.add_argument("--workload", default="unet3d", choices=["unet3d", "svd", "rgba"],
                   help="unet3d: BASELINE configs[1] (the metric of record); svd: configs[3]; rgba: configs[4]")
    a = p.parse_args()
    if a.workload == "rgba" and a.size == 512:
        a.size = 384                                          # configs[4]: 16 frames x 384 x 384 (48 x 48 latents)
    return a


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


def cpu_baseline(reps=3):
    """The oracle (CPU restatement of the reference: kind "port") on the host cores, a REAL measurement of a whole
    workload, no extrapolation: BASELINE.json configs[0] - 8 frames x 256x256 (latents [2,4,8,32,32] with CFG, 9 frames
    inside; 5.495 TFLOP per step) - one full UNet3D forward, median of `reps`.  `value` is that
    configuration's steps/s.  The one full 16 frames x 512x512 forward (44.262 TFLOP) measured offline on the GPU box's
    host cores is quoted from profiles/r02_cpu_baseline.json when present (`full_config_*`)."""
    import oracle
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    # timing does not depend on the weight values: skip the 1.4 G-element default init, fill with small uniforms
    with torch.device("meta"):
        net = oracle.UNet3DConditionModel(motion_mask=True, motion_strength=True)
    net = net.to_empty(device="cpu").eval()
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if p_.dim() == 1 and "norm" in n_ and n_.endswith("weight"):
                p_.fill_(1.0)
            else:
                p_.uniform_(-0.03, 0.03, generator=g)
    g = torch.Generator().manual_seed(1234)
    r = lambda *s: torch.randn(*s, generator=g)
    lat, frames = 32, 8
    x, c, txt = r(2, 4, frames, lat, lat), r(2, 4, 1, lat, lat), r(2, 77, 1024)
    m = torch.zeros(1, 1, 1, lat, lat)
    m[..., 8:24, 8:24] = 1
    ts = []
    with torch.no_grad():
        for _ in range(reps):
            t0 = time.perf_counter()
            y = net(x, 500, txt, c, m, motion=torch.tensor([3.0])).sample
            ts.append(time.perf_counter() - t0)
    assert torch.isfinite(y).all()
    ts.sort()
    dt = ts[len(ts) // 2]
    out = dict(value=round(1.0 / dt, 5), unit="steps/s", cores=cores, kind="port",
               sample=f"oracle UNet3D forward fp32 on {cores} threads, BASELINE configs[0]: 8 frames x 256x256 (latents "
                      f"[2,4,8,32,32] with CFG, 9 frames inside, 5.495 TFLOP), whole forward, median of {reps}: {dt:.2f} s "
                      f"(all runs {', '.join('%.2f' % t for t in ts)} s); value = steps/s OF THAT CONFIG, not of the 16fx512x512 metric",
               tflops=round(5.495 / dt, 3))
    full = os.path.join(ROOT, "profiles", "r02_cpu_baseline.json")
    if os.path.exists(full):
        rec = json.load(open(full))
        out["full_config_steps_per_s"] = rec.get("steps_per_s")
        out["full_config_note"] = (f"one full 16fx512x512 CFG forward (44.262 TFLOP) measured offline: {rec.get('seconds_per_step'):.1f} s "
                                   f"on {rec.get('cores')} threads of {rec.get('cpu')} (profiles/r02_cpu_baseline.json)")
    return out


def contraction_roofline(trace, gemm_breakdown, flop_step, step_ms):
    """Re-time the recorded contraction calls of ONE step back to back on the current stream (the stream the kernels run on)
    between two events, and account for them: FLOP and ALGORITHMIC bytes from the descriptors (unique input + weights +
    output (+ residual) of every call, what a perfect kernel moves once), kernel launches from the library's own plan
    (aa_conv_gemm_launch_count minus aa_conv_gemm_reduce_launches: main launch + split-off last round; the split-K reduce launches that
    remain - K splits of the hand-scheduled tiles finish inside the kernel - are reported apart), measured HBM bytes from the last committed
    PMC run of the same command (profiles/r0N_traffic_pmc.json: FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate rocprofv3 passes -
    bench.py cannot profile itself)."""
    import ctypes as C
    from animate_anything_amd import _lib
    lib = _lib.get()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    flops = sum(2.0 * d.n_img * d.h_out * d.w_out * d.n_out * d.kh * d.kw * (d.c0 + d.c1) for d, _ in trace)
    # ONE definition: contraction-kernel launches and split-K reduce launches are counted apart; `avg_kernel_launch_us`
    # and the PMC bytes per launch refer to the contraction kernels only
    reduces = sum(int(lib.aa_conv_gemm_reduce_launches(C.byref(d))) for d, _ in trace)
    # calls the LDS-DMA family cannot carry (channels not a multiple of 64, output rows not 16-byte pieces: conv_in2) run the generic gather
    # kernel - one launch each, counted apart: `launches` is what the PMC / rocprof records list under the contraction family
    generic = sum(1 for d, _ in trace if (d.c0 + d.c1) % 64 or d.c0 % 64 or (d.n_out // 2 if d.geglu else d.n_out) % 8 or d.out_dtype != d.dtype)
    launches = sum(int(lib.aa_conv_gemm_launch_count(C.byref(d))) for d, _ in trace) - reduces - generic
    alg_bytes = 0.0
    for d, _ in trace:
        m = d.n_img * d.h_out * d.w_out
        n_cols = d.n_out // 2 if d.geglu else d.n_out
        alg_bytes += 2.0 * (d.n_img * d.h_in * d.w_in * (d.c0 + d.c1) + d.n_out * d.kh * d.kw * (d.c0 + d.c1) + m * n_cols
                            + (m * n_cols if d.residual else 0))
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
    if gemm_breakdown:
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
        with open(gemm_breakdown, "w") as f:
            f.write("M K N kind flags | launches total_ms TFLOP/s tile share\n")
            tot = sum(v[1] for v in groups.values())
            for key, v in sorted(groups.items(), key=lambda kv: -kv[1][1]):
                f.write(f"{key[0]:7d} {key[1]:6d} {key[2]:6d} {key[3]:6s} {key[4]:5s}{key[5]:3s} | {v[0]:3d} {v[1]:8.3f} "
                        f"{v[2] / (v[1] * 1e-3) / 1e12:7.1f} {v[3]:>4s} {v[1] / tot * 100:5.1f}%\n")
    ach = flops / (gemm_ms * 1e-3) / 1e12
    # bench.py cannot profile itself: the PMC bytes come from the newest committed run of scripts/pmc_traffic.sh - and only count
    # when that run was of THIS library and THIS launch plan (a kernel change without a new PMC run used to report
    # stale bytes silently).  The record carries the sha256 of the libaa_mi355.so it profiled.
    traffic, tname, t_launches, stale = pick_traffic_record(os.path.join(ROOT, "profiles"), library_id(), launches)
    traffic_step = traffic * t_launches if traffic and t_launches else None
    return {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
            "traffic_unit": (f"HBM bytes per KERNEL LAUNCH of the contraction kernels, measured (PMC FETCH_SIZE x 2 + WRITE_SIZE in separate "
                             f"rocprofv3 passes, profiles/{tname}: same library sources {library_id()}, {t_launches} such launches per step)") if traffic
                            else f"null: no PMC measurement of this library ({stale or 'no profiles/r*_traffic_pmc.json'}); run scripts/pmc_traffic.sh",
            "library_source_sha256_16": library_id(),
            "kernel": "contraction kernels: aa::conv_gemm_dma_kernel / conv3x3_slab_kernel / conv_gemm_x_kernel (LDS-DMA implicit-GEMM conv / linear), all instances",
            "contraction_launches_per_step": launches, "reduce_launches_per_step": reduces, "generic_kernel_launches_per_step": generic,
            "aa_conv_gemm_calls_per_step": len(trace),
            "avg_kernel_launch_us": round(gemm_ms * 1e3 / max(launches, 1), 2),
            "flop_per_step_in_kernel": flops, "kernel_ms_per_step": round(gemm_ms, 3),
            "algorithmic_bytes_per_step": round(alg_bytes), "traffic_bytes_per_step": traffic_step,
            "traffic_over_algorithmic": round(traffic_step / alg_bytes, 3) if traffic_step else None,
            "whole_step_frac_of_peak": round(flop_step / (step_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)}


def write_launch_list(path, unet, one_step):
    """One eager step's aa_conv_gemm calls in launch order: per call the shape, how many contraction-kernel launches it makes (main launch +
    split-off last round; generic-kernel and split-K reduce launches listed apart) and its algorithmic bytes (unique input + weights +
    output (+ residual)) - the per-dispatch rows of a rocprofv3 PMC pass of the same command are joined with this list
    (scripts/pmc_traffic_report.py: measured / algorithmic bytes PER SHAPE)."""
    import ctypes as C
    from animate_anything_amd import _lib, ops
    lib = _lib.get()
    unet.enable_graph(False)
    ops.TRACE = []
    with torch.no_grad():
        one_step()
    trace, ops.TRACE = ops.TRACE, None
    torch.cuda.synchronize()
    rows = []
    for d, _ in trace:
        m = d.n_img * d.h_out * d.w_out
        n_cols = d.n_out // 2 if d.geglu else d.n_out
        k = d.kh * d.kw * (d.c0 + d.c1)
        generic = int(bool((d.c0 + d.c1) % 64 or d.c0 % 64 or n_cols % 8 or d.out_dtype != d.dtype))
        reduces = int(lib.aa_conv_gemm_reduce_launches(C.byref(d)))
        rows.append({"M": m, "K": k, "N": d.n_out, "kind": f"{d.kh}x{d.kw}s{d.stride}", "geglu": int(bool(d.geglu)), "two_source": int(d.c1 != 0),
                     "residual": int(bool(d.residual)), "ln_fold": int(bool(d.ln_stats)), "tile": int(d.tile), "k_splits": int(d.k_splits),
                     "launches": int(lib.aa_conv_gemm_launch_count(C.byref(d))) - reduces - generic, "reduce_launches": reduces, "generic": generic,
                     "algorithmic_bytes": 2 * (d.n_img * d.h_in * d.w_in * (d.c0 + d.c1) + d.n_out * k + m * n_cols + (m * n_cols if d.residual else 0)),
                     "flop": 2.0 * m * k * d.n_out})
    with open(path, "w") as f:
        json.dump(rows, f)


def pick_traffic_record(profiles_dir, lib_id, launches):
    """The newest committed PMC record (profiles/r*_traffic_pmc.json, scripts/pmc_traffic.sh) - quoted only if it was measured on THESE library
    sources and on the same number of contraction launches per step.  Returns (hbm bytes per launch | None, file name | None, launches per
    step of the record | None, why not | None)."""
    import glob
    for tpath in sorted(glob.glob(os.path.join(profiles_dir, "r*_traffic_pmc.json")), reverse=True):
        rec = json.load(open(tpath))
        fam = rec.get("contraction_kernels") or rec.get("conv_gemm_dma_kernel") or {}
        tname = os.path.basename(tpath)
        if rec.get("library_source_sha256_16") != lib_id:
            return None, tname, None, (f"profiles/{tname} was measured on library sources {rec.get('library_source_sha256_16', '(unrecorded)')}, "
                                       f"these are {lib_id}")
        if fam.get("launches_per_step") != launches:
            return None, tname, None, f"profiles/{tname} had {fam.get('launches_per_step')} contraction launches per step, this run has {launches}"
        return fam.get("hbm_bytes_per_launch"), tname, fam.get("launches_per_step"), None
    return None, None, None, None


_LIB_ID = None


def library_id():
    """sha256[:16] over the SOURCES of libaa_mi355.so (animate_anything_amd.build.source_id: what a committed PMC record must name to be
    quoted - a rebuild of the same sources elsewhere is the same library, an edited kernel is not)."""
    global _LIB_ID
    if _LIB_ID is None:
        from animate_anything_amd import build
        _LIB_ID = build.source_id()
    return _LIB_ID


def vae_timing(dtype, device, ms_step):
    """SURVEY section 8(d) "VAE encode + decode ms per clip": the SD AutoencoderKL (seeded random weights) around the denoising loop of
    one clip - encode of the ONE conditioning frame at 512x512 (reference utils/common.py:12-20 via train.py:766-770) and decode
    of the 16 generated frames (models/pipeline.py:200) - once per clip, not per step, so reported beside `value`, not inside it.
    FLOP: BASELINE.md section 2 (encode 1.117 TFLOP / frame, decode 2.515 TFLOP / frame)."""
    from animate_anything_amd.vae import AutoencoderKL
    torch.manual_seed(0)
    with torch.device(device):
        vae = AutoencoderKL()
    vae = vae.to(dtype).eval()
    g = torch.Generator(device=device).manual_seed(7)
    img = (torch.rand(1, 3, 512, 512, generator=g, device=device) * 2 - 1).to(dtype)
    z = torch.randn(16, 4, 64, 64, generator=g, device=device).to(dtype)

    def timed(fn, reps=3):
        y = fn()                                            # (first call: weight packing, tile choices)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            y = fn()
        torch.cuda.synchronize()
        assert torch.isfinite(y).all()
        return (time.perf_counter() - t0) / reps * 1e3

    with torch.no_grad():
        enc = timed(lambda: vae.encode(img).latent_dist.mode())
        dec = timed(lambda: vae.decode(z).sample)
    return {"encode_1x512x512_ms": round(enc, 3), "decode_16x512x512_ms": round(dec, 3),
            "encode_tflops": round(1.117 / enc * 1e3, 1), "decode_tflops": round(16 * 2.515 / dec * 1e3, 1),
            "per_clip_ms": round(enc + dec, 3),
            "note": "AutoencoderKL (SD VAE architecture, seeded random init), once per clip: encode of the conditioning frame + decode of "
                    "the 16 generated frames; = %.2f denoising steps of this run" % ((enc + dec) / ms_step)}


def respawn_under_torchrun(a, script):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start N ranks on this node, one per GPU."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def rgba_addons(dtype, device, frames, size):
    """The transparent-VAE add-ons of BASELINE configs[4] (reference models/layerdiffuse_VAE.py:17-177, call sites
    train_transparent_i2v_stage2.py:400-426 and models/pipeline_stage2.py:290-318), once per clip: the latent-offset encoder on
    the RGBA conditioning frame and the UNet384 alpha decoder on the `frames` decoded frames (seeded random weights)."""
    from animate_anything_amd import layerdiffuse as P
    torch.manual_seed(3)
    with torch.device(device):
        enc, dec = P.LatentTransparencyOffsetEncoder().to(dtype).eval(), P.UNet384().to(dtype).eval()
    g = torch.Generator(device=device).manual_seed(5)
    rgba = torch.rand(1, 4, size, size, generator=g, device=device).to(dtype) * 2 - 1
    pix = torch.rand(frames, 3, size, size, generator=g, device=device).to(dtype) * 2 - 1
    lat = torch.randn(frames, 4, size // 8, size // 8, generator=g, device=device).to(dtype)
    res = {}
    with torch.no_grad():
        for name, fn in (("offset_encoder_ms", lambda: enc(rgba)), (f"alpha_decoder_{frames}_frames_ms", lambda: dec(pix, lat))):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                y = fn()
            torch.cuda.synchronize()
            res[name] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
            assert torch.isfinite(y).all()
    return res


def cpu_baseline_svd(frames):
    """Oracle (oracle/svd.py, fp32) forward of the full SVD UNet on a BOUNDED sample of the workload: the real architecture and
    frame count on a 24 x 32 latent grid (1/12 of the 72 x 128 tokens), CFG batch 2.  `value` = steps/s of that sample."""
    import oracle.svd as O
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    with torch.device("meta"):
        net = O.UNetSpatioTemporalConditionModel(in_channels=9, num_frames=frames)
    net = net.to_empty(device="cpu").eval()
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if p_.dim() == 1 and "norm" in n_ and n_.endswith("weight"):
                p_.fill_(1.0)
            elif n_.endswith("mix_factor"):
                p_.fill_(0.3)
            else:
                p_.uniform_(-0.03, 0.03, generator=g)
    h, w = 24, 32
    x = torch.randn(2, frames, 9, h, w, generator=g)
    emb = torch.randn(2, 1, 1024, generator=g)
    ids = torch.tensor([[6.0, 127.0, 0.02]]).repeat(2, 1)
    ts = []
    with torch.no_grad():
        for _ in range(2):
            t0 = time.perf_counter()
            y = net(x, 1.3, emb, ids).sample
            ts.append(time.perf_counter() - t0)
    assert torch.isfinite(y).all()
    dt = min(ts)
    return dict(value=round(1.0 / dt, 5), unit="steps/s", cores=cores, kind="port",
                sample=f"oracle SVD UNet forward fp32 on {cores} threads: full architecture, {frames} frames, 24 x 32 latents (1/12 of the "
                       f"72 x 128 grid), CFG batch 2: {dt:.2f} s (runs {', '.join('%.2f' % t for t in ts)} s); value = steps/s OF THAT SAMPLE")


def run_svd(a, rank, world, device):
    """BASELINE configs[3]: Stable-Video-Diffusion path, 14 frames x 576 x 1024, bs=1 (reference models/pipeline.py:413-451 /
    train_svd.py --eval).  One step = input assembly + UNetSpatioTemporalConditionModel forward on the CFG-doubled batch + per-frame
    guidance + Euler update.  Same JSON schema as the UNet3D workload."""
    from animate_anything_amd import ops
    from animate_anything_amd.schedulers import EulerDiscreteScheduler
    from animate_anything_amd.svd_pipeline import StableVideoDiffusionPipeline
    from animate_anything_amd.svd_unet import UNetSpatioTemporalConditionModel
    assert world == 1, "the SVD workload is a single-GPU side measurement"
    dtype = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    frames, height, width = (14, 576, 1024) if (a.frames, a.size) == (16, 512) else (a.frames, a.size, a.size)
    torch.manual_seed(0)
    with torch.device(device):
        unet = UNetSpatioTemporalConditionModel(in_channels=9, num_frames=frames).to(dtype).eval()
    with torch.no_grad():
        for n_, p_ in unet.named_parameters():
            if n_.endswith("mix_factor"):
                p_.fill_(0.3)
    if a.no_tile_cache:
        os.environ["AA_NO_TILE_CACHE"] = "1"
    if a.tile_cache and os.path.exists(a.tile_cache):
        ops.load_tile_cache(a.tile_cache)
    if not a.no_graph:
        unet.enable_graph()
    pipe = StableVideoDiffusionPipeline(None, None, unet, EulerDiscreteScheduler())
    g = torch.Generator(device=device).manual_seed(1234)
    h, w, f = height // 8, width // 8, frames
    latents = torch.randn(1, f, 4, h, w, generator=g, device=device)
    cond = torch.randn(1, 1, 4, h, w, generator=g, device=device).repeat(1, f, 1, 1, 1).to(dtype)
    cond = torch.cat([torch.zeros_like(cond), cond])
    emb = torch.randn(1, 1, 1024, generator=g, device=device).to(dtype)
    emb = torch.cat([torch.zeros_like(emb), emb])
    mask = torch.zeros(2, f, 1, h, w, device=device, dtype=dtype)
    mask[..., h // 4: h - h // 4, w // 4: w - w // 4] = 1
    ids = torch.tensor([[6.0, 127.0, 0.02]], device=device).repeat(2, 1)
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
    ms = dt / a.steps * 1e3
    # FLOP census of one step from the launches themselves: contractions from their descriptors, attention analytically
    unet.enable_graph(False)
    ops.TRACE = []
    att = []
    real_attention = ops.attention

    def counting_attention(q, q0, k, k0, v, v0, heads, n_outer, n_inner, q_len, kv_len, *r, **kw):
        att.append(4.0 * n_outer * n_inner * heads * q_len * kv_len * kw.get("head_dim", 64))
        return real_attention(q, q0, k, k0, v, v0, heads, n_outer, n_inner, q_len, kv_len, *r, **kw)

    # (the fused kernels that are not aa_conv_gemm calls: counted where they are called)
    real_rows, real_ff, real_seq = ops.linear_rows, ops.ff_fused, ops.seq_self_attention

    def counting_rows(x, pk, *r, **kw):
        att.append(2.0 * x.shape[0] * x.shape[1] * pk.n_out)
        return real_rows(x, pk, *r, **kw)

    def counting_ff(x, pk, *r, **kw):
        att.append(2.0 * x.shape[0] * x.shape[1] * 13 * x.shape[1])                 # GEGLU 8 C + ff-out 4 C + proj_out C outputs per row
        return real_ff(x, pk, *r, **kw)

    def counting_seq(x, pk, clips, hw, frames, *r, **kw):
        att.append(2.0 * x.shape[0] * x.shape[1] * 3 * x.shape[1] + 4.0 * x.shape[0] * frames * x.shape[1])
        return real_seq(x, pk, clips, hw, frames, *r, **kw)

    ops.attention, ops.linear_rows, ops.ff_fused, ops.seq_self_attention = counting_attention, counting_rows, counting_ff, counting_seq
    with torch.no_grad():
        run(ts[:1], latents)
    ops.attention, ops.linear_rows, ops.ff_fused, ops.seq_self_attention = real_attention, real_rows, real_ff, real_seq
    trace, ops.TRACE = ops.TRACE, None
    torch.cuda.synchronize()
    gemm_flop = sum(2.0 * d.n_img * d.h_out * d.w_out * d.n_out * d.kh * d.kw * (d.c0 + d.c1) for d, _ in trace)
    flop_step = gemm_flop + sum(att)
    out = {"metric": f"SVD denoising steps/sec @{f}fx{height}x{width} bs=1 (BASELINE configs[3])", "value": round(a.steps / dt, 4), "unit": "steps/s",
           "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
           "config": {"workload": f"stable-video-diffusion-img2vid UNetSpatioTemporalConditionModel, 9 input channels "
                                  f"({sum(p_.numel() for p_ in unet.parameters()) / 1e6:.0f}M params, seeded random init), {f} frames x "
                                  f"{height}x{width}, CFG batch 2, 1 context token, Euler step, hipGraph={'off' if a.no_graph else 'on'}"},
           "flop_per_step_executed": flop_step, "attention_and_fused_kernel_flop_per_step": sum(att), "tflops_per_gpu": round(flop_step / (ms * 1e-3) / 1e12, 1),
           "autotuned_signatures": int(ops.AUTOTUNE_EVENTS)}
    if not a.no_roofline:
        out["roofline"] = contraction_roofline(trace, a.gemm_breakdown, flop_step, ms)
        out["roofline"]["traffic"] = out["roofline"]["traffic_bytes_per_step"] = out["roofline"]["traffic_over_algorithmic"] = None   # (the committed PMC run is of the UNet3D command)
    if not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_svd(frames)
    print(json.dumps(out), flush=True)


def main(selftest_library=None, script=None):
    """`selftest_library` / `script`: tests/bench_selftest.py drives this script's N > 1 control flow on CPU ranks (gloo, the test
    suite's CPU build of the kernels, a toy architecture; the line it prints says "selftest" and is not a measurement) by passing
    a context manager that swaps the library and its own path for the torchrun respawn.  bench.py itself knows no CPU path."""
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        respawn_under_torchrun(a, script or os.path.abspath(__file__))
    if selftest_library is not None:
        with selftest_library():
            return bench(a, selftest=True)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    return bench(a)


def bench(a, selftest=False):
    from animate_anything_amd import distributed as D
    rank, world, device = D.init("gloo" if selftest else "nccl")
    sync = (lambda: None) if selftest else torch.cuda.synchronize
    pinned = D.pin_to_gpu_numa_node(device.index or 0) if world > 1 else None      # ranks stay on their GPU's NUMA node
    if a.workload == "svd":
        return run_svd(a, rank, world, device)
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if world > 1:
        import torch.distributed as dist
        n_ranks = torch.ones(1, device=device)
        dist.all_reduce(n_ranks)                              # RCCL sanity: every rank is really there
        assert int(n_ranks.item()) == world
    dtype = torch.float16 if a.dtype == "fp16" else torch.bfloat16

    from animate_anything_amd import ops
    from animate_anything_amd.pipeline import LatentToVideoPipeline
    from animate_anything_amd.schedulers import DPMSolverMultistepScheduler

    lat = a.size // 8
    if selftest:
        from util import TINY_UNET
        from animate_anything_amd.unet3d import UNet3DConditionModel
        torch.manual_seed(0)
        unet = UNet3DConditionModel(**TINY_UNET).eval()
        with torch.no_grad():
            for p_ in unet.parameters():
                if p_.abs().max() == 0:
                    p_.normal_(0.0, 0.02)
        unet = unet.to(dtype)
        a.no_graph = a.no_roofline = a.no_cpu_baseline = True
    else:
        unet = build_unet(dtype, device)
    pipe = LatentToVideoPipeline(vae=None, unet=unet, scheduler=DPMSolverMultistepScheduler())
    num_clips = world                                                             # weak scaling: one clip per GPU
    mine = D.clip_indices(num_clips, rank, world)                                 # == [rank]
    inp = synthetic_inputs(a.frames, lat, dtype, device, seed=D.clip_seed(1234, mine[0]))
    if selftest:                                                                  # toy geometry: 2 frames of 5 x 6 latents, 9 text tokens of 64
        g = torch.Generator().manual_seed(D.clip_seed(1234, mine[0]))
        r = lambda *sh: torch.randn(*sh, generator=g)
        m = torch.zeros(1, 1, 1, 5, 6)
        m[..., 1:4, 2:5] = 1
        inp = dict(latents=r(1, 4, 2, 5, 6), cond=r(1, 4, 1, 5, 6).to(dtype), mask=m.to(dtype), text=r(1, 9, 64).to(dtype), neg=r(1, 9, 64).to(dtype))
    embeds = torch.cat([inp["neg"], inp["text"]])
    total = a.warmup + a.steps
    pipe.scheduler.set_timesteps(max(total, 2))
    ts = [int(t) for t in pipe.scheduler.timesteps][:total]
    if not a.no_graph:
        unet.enable_graph()
    if a.no_tile_cache:
        os.environ["AA_NO_TILE_CACHE"] = "1"
    if a.tile_cache and os.path.exists(a.tile_cache):
        ops.load_tile_cache(a.tile_cache)

    # Headline = the STRICT step: both guidance halves run the whole UNet, as the reference does (44.262 TFLOP).  The product's
    # default computes the text-independent prefix once per pair (the same arithmetic, 1.64 TFLOP less; latents equal to the strict form
    # within the fp16 noise floor, not bit for bit - `max_abs_latent_difference_after_2_steps` below): timed too, reported
    # separately as `cfg_shared_prefix` (or as the headline with --cfg-shared-prefix).
    pipe.cfg_shared_prefix = bool(a.cfg_shared_prefix)

    def run(tsteps, x):
        return pipe.denoise(x, embeds, inp["cond"], inp["mask"], [3.0], tsteps, 9.0)

    def barrier():
        if world > 1:
            dist.barrier()
        sync()

    with torch.no_grad():
        x = run(ts[: a.warmup], inp["latents"]) if a.warmup else inp["latents"]
        if a.tile_cache and rank == 0:
            ops.save_tile_cache(a.tile_cache)
        barrier()
        t0 = time.perf_counter()
        x = run(ts[a.warmup:], x)
        gathered = D.gather_clips(x.to(dtype).contiguous(), num_clips, rank, world)   # the ONE collective (RCCL all-gather)
        barrier()
        dt = time.perf_counter() - t0
    per_rank_ms = [dt / a.steps * 1e3]
    if world > 1:
        tall = [torch.zeros(1, device=device) for _ in range(world)]
        dist.all_gather(tall, torch.tensor([dt], device=device))
        per_rank_ms = [round(t.item() / a.steps * 1e3, 3) for t in tall]
        dt = max(t.item() for t in tall)
    assert gathered.shape[0] == num_clips
    assert torch.isfinite(x).all() and torch.isfinite(gathered).all(), "non-finite latents"

    # the other form of the step (shared prefix on <-> off), same number of steps, same protocol
    other = None
    if not a.no_other_form:
        pipe.cfg_shared_prefix = not a.cfg_shared_prefix
        with torch.no_grad():
            xo = run(ts[: max(a.warmup, 1)], inp["latents"])
            barrier()
            t1 = time.perf_counter()
            xo = run(ts[a.warmup:], xo)
            barrier()
            dto = time.perf_counter() - t1
        if world > 1:
            tmx = torch.tensor([dto], device=device)
            dist.all_reduce(tmx, op=dist.ReduceOp.MAX)
            dto = tmx.item()
        other = {"cfg_shared_prefix": not a.cfg_shared_prefix, "value": round(world * a.steps / dto, 4), "ms_per_step": round(dto / a.steps * 1e3, 3),
                 "note": "same step with the text-independent UNet prefix (conv_in, transformer_in, first resnet / temporal conv / spatial "
                         "self-attention) computed once per guidance pair instead of twice: the same arithmetic per element on other tiles (other "
                         "summation orders), 1.641 TFLOP less executed; the measured difference of the latents after two steps "
                         "is max_abs_latent_difference_after_2_steps (fp16 rounding noise amplified by guidance 9 on random weights; "
                         "tests/test_gpu_fullsize.py bounds it by the measured noise floor of the strict form and checks both forms against the oracle)"}
        # cross-check of the two forms where rounding differences have not been amplified yet by the (random-weight, guidance 9)
        # dynamics: two steps from the same initial latents
        with torch.no_grad():
            two = []
            for flag in (False, True):
                pipe.cfg_shared_prefix = flag
                two.append(run(ts[:2], inp["latents"]).float())
        other["max_abs_latent_difference_after_2_steps"] = (two[0] - two[1]).abs().max().item()
        other["latent_abs_max_after_2_steps"] = two[0].abs().max().item()
        pipe.cfg_shared_prefix = bool(a.cfg_shared_prefix)
    flop_full = FLOP_TABLE.get((a.frames, lat), FLOP_PER_STEP * (a.frames + 1) / 17 * (lat / 64) ** 2)
    # `flop_per_step_reference` = the census of the reference's arithmetic; `flop_per_step_executed` (what the rates below are computed from) leaves
    # out what the product does not run: the 5 of 9 upsampler taps, and the text-independent prefix of one guidance half in the shared-prefix form
    flop_step = flop_full - UPSAMPLE_FLOP_NOT_EXECUTED * flop_full / FLOP_PER_STEP - (1.641e12 * flop_full / FLOP_PER_STEP if a.cfg_shared_prefix else 0.0)
    ms_step = dt / a.steps * 1e3
    value = world * a.steps / dt
    if world > 1 and ops.AUTOTUNE_EVENTS and rank == 0:
        print(f"bench.py: WARNING {ops.AUTOTUNE_EVENTS} contraction signatures were autotuned inside warm-up on rank 0 (the committed "
              "tile cache animate_anything_amd/tile_cache_gfx950.json does not cover this library / table version)", file=sys.stderr, flush=True)
    metric = f"UNet3D denoising steps/sec @{a.frames}fx{a.size}x{a.size} bs=1" + (" (layerdiffuse RGBA path, BASELINE configs[4])" if a.workload == "rgba" else "")
    out = {
        "metric": metric, "value": round(value, 4), "unit": "steps/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        **({"selftest": "CPU ranks over gloo on the test suite's CPU build of the kernels, toy architecture: control flow only, NOT a measurement"} if selftest else {}),
        "config": {"workload": f"animate_anything_512_v1.02 UNet3D (1413M params, seeded random init), "
                               f"{a.frames} frames x {a.size}x{a.size}, pipeline bs=1 per GPU (CFG batch 2, 17 frames "
                               f"inside), DPM-Solver++ step, hipGraph={'off' if a.no_graph else 'on'}",
                   "clips_per_gpu": 1, "clips": num_clips, "parallelism": f"clip-sharded x{world}",
                   "collective": "none in the data path; one all_gather_into_tensor of the final latents (RCCL)" if world > 1 else "none"},
        "per_rank_ms_per_step": per_rank_ms,
        "tflops_per_gpu": round(flop_step * (a.steps / dt) / 1e12, 2),
        "flop_per_step_executed": flop_step, "flop_per_step_reference": flop_full, "cfg_shared_prefix": bool(a.cfg_shared_prefix), "other_form": other,
    }

    if rank == 0 and not a.no_roofline:
        # dominant kernel in isolation: record one eager step's contraction calls, replay them back to back
        unet.enable_graph(False)
        ops.TRACE = []
        with torch.no_grad():
            run(ts[:1], inp["latents"])
        trace, ops.TRACE = ops.TRACE, None
        sync()
        out["roofline"] = contraction_roofline(trace, a.gemm_breakdown, flop_step, ms_step)
        if a.workload != "unet3d" or (a.frames, a.size, a.dtype) != (16, 512, "fp16"):      # the committed PMC run is of the default command
            out["roofline"]["traffic"] = out["roofline"]["traffic_bytes_per_step"] = out["roofline"]["traffic_over_algorithmic"] = None
    out["autotuned_signatures"] = int(ops.AUTOTUNE_EVENTS)      # 0 = every contraction signature came from the committed tile cache
    if rank == 0 and a.launch_list:
        write_launch_list(a.launch_list, unet, lambda: run(ts[:1], inp["latents"]))
    if pinned:
        out["cpu_affinity"] = f"{len(pinned)} CPUs of the GPU's NUMA node"
    if a.workload == "rgba" and rank == 0:
        out["rgba_addons"] = rgba_addons(dtype, device, a.frames, a.size)
    if a.workload == "unet3d" and rank == 0 and world == 1 and not selftest and not a.no_vae:
        out["vae"] = vae_timing(dtype, device, ms_step)
        out["autotuned_signatures_vae"] = int(ops.AUTOTUNE_EVENTS) - out["autotuned_signatures"]
    if a.tile_cache and rank == 0:
        ops.save_tile_cache(a.tile_cache)                       # (again: the other guidance form and the eager trace add signatures)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
