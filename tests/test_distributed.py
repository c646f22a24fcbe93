"""N>1 path on CPU: two gloo ranks shard a batch of clips, each runs the REAL denoising loop
(`LatentToVideoPipeline.denoise`: product UNet3D on the SIMT emulator + fused CFG/DPM-Solver++ step) on its own clips with
per-clip seeds and the final latents are all-gathered in clip order - the result must equal the single-process run
(the GPU path swaps gloo for RCCL and the emulator for libaa_mi355.so; everything else is the same code:
animate_anything_amd/distributed.py, used by bench.py and eval.main_eval)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from animate_anything_amd import distributed as D

HERE = os.path.dirname(os.path.abspath(__file__))


def _pipeline():
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import build_emu
    from animate_anything_amd import _lib
    from animate_anything_amd.pipeline import LatentToVideoPipeline
    from animate_anything_amd.schedulers import DPMSolverMultistepScheduler
    from animate_anything_amd.unet3d import UNet3DConditionModel
    from util import TINY_UNET
    lib = _lib.bind(build_emu.build())
    torch.manual_seed(0)
    unet = UNet3DConditionModel(**TINY_UNET).eval()
    with torch.no_grad():
        for p_ in unet.parameters():
            if p_.abs().max() == 0:
                p_.normal_(0.0, 0.02)
    pipe = LatentToVideoPipeline(vae=None, unet=unet.half(), scheduler=DPMSolverMultistepScheduler())
    return lib, pipe


def _denoise_clip(pipe, clip_index, seed, steps=2):
    g = torch.Generator().manual_seed(D.clip_seed(seed, clip_index))
    r = lambda *s: torch.randn(*s, generator=g)
    lat, cond, pos, neg = r(1, 4, 2, 5, 6), r(1, 4, 1, 5, 6), r(1, 9, 64), r(1, 9, 64)
    mask = torch.zeros(1, 1, 1, 5, 6)
    mask[..., 1:4, 2:5] = 1
    pipe.scheduler.set_timesteps(steps)
    ts = [int(t) for t in pipe.scheduler.timesteps]
    with torch.no_grad():
        x = pipe.denoise(lat, torch.cat([neg, pos]).half(), cond.half(), mask.half(), [3.0], ts, 9.0)
    return x[0]


def _run_clips(indices, seed):
    from animate_anything_amd import _lib
    lib, pipe = _pipeline()
    with _lib.use_library(lib, host_pointers=True):
        return [_denoise_clip(pipe, i, seed) for i in indices]


def _worker(rank, world, port, num_clips, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AA_EMU_THREADS="4")
    torch.set_num_threads(2)
    r, w, dev = D.init("gloo")
    mine = D.clip_indices(num_clips, r, w)
    outs = _run_clips(mine, 100)
    local = torch.stack(outs) if outs else torch.zeros(0, 4, 2, 5, 6)
    full = D.gather_clips(local, num_clips, r, w)
    q.put((rank, full))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_clip_sharding_of_the_real_denoising_loop_matches_single_rank():
    num_clips = 3                                          # uneven: rank 0 owns clips 0 and 2, rank 1 owns clip 1
    want = torch.stack(_run_clips(range(num_clips), 100))
    assert torch.isfinite(want).all() and (want[0] - want[1]).abs().max() > 1e-3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, num_clips, q)) for r in range(2)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=300) for _ in range(2))
    [p.join(60) for p in procs]
    for r in range(2):
        assert torch.equal(got[r], want)                   # bit-identical: ownership and rank count do not change a clip


def test_single_rank_is_identity():
    x = torch.randn(3, 2)
    assert D.gather_clips(x, 3, 0, 1) is x
    assert D.clip_indices(5, 1, 2) == [1, 3]
    assert D.clip_seed(7, 3) == 10


def _pair_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AA_EMU_THREADS="4")
    torch.set_num_threads(2)
    r, w, dev = D.init("gloo")
    pair, role, group = D.guidance_pair(r, w)
    from animate_anything_amd import _lib
    lib, pipe = _pipeline()
    pipe.guidance_group = (role, group)
    with _lib.use_library(lib, host_pointers=True):
        out = _denoise_clip(pipe, 0, 100)
    q.put((rank, pair, role, out))
    dist.barrier()
    dist.destroy_process_group()


def test_guidance_parallel_pair_matches_one_rank():
    """Latency mode: the unconditional and the text half of one clip's guidance batch on two ranks, one all-gather of the
    UNet outputs per step (gloo here, RCCL on the GPUs) - same latents as the single-rank loop on both ranks."""
    want = _run_clips([0], 100)[0]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + ((os.getpid() + 137) % 500)
    procs = [ctx.Process(target=_pair_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got = [q.get(timeout=300) for _ in range(2)]
    [p.join(60) for p in procs]
    assert sorted((g[0], g[1], g[2]) for g in got) == [(0, 0, 0), (1, 0, 1)]
    assert torch.equal(got[0][3], got[1][3])                       # both ranks hold the same latents
    err = (got[0][3].float() - want.float()).abs().max() / want.float().abs().max()
    assert err < 2e-3, err                                         # (batch 1 vs batch 2 contractions may pick other tile plans)
