"""N>1 path on CPU: two gloo ranks shard a batch of clips, run a deterministic per-clip stand-in for the
denoising loop and all-gather the results in clip order (the GPU path swaps gloo for RCCL)."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from animate_anything_amd import distributed as D


def _fake_denoise(clip_index, seed):
    g = torch.Generator().manual_seed(D.clip_seed(seed, clip_index))
    return torch.randn(4, 3, 5, 5, generator=g)


def _worker(rank, world, port, num_clips, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, dev = D.init("gloo")
    mine = D.clip_indices(num_clips, r, w)
    local = torch.stack([_fake_denoise(i, 100) for i in mine]) if mine else torch.zeros(0, 4, 3, 5, 5)
    full = D.gather_clips(local, num_clips, r, w)
    q.put((rank, full))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_clip_sharding_matches_single_rank():
    for num_clips in (4, 5):
        want = torch.stack([_fake_denoise(i, 100) for i in range(num_clips)])
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = 29500 + (os.getpid() % 500) + num_clips
        procs = [ctx.Process(target=_worker, args=(r, 2, port, num_clips, q)) for r in range(2)]
        [p.start() for p in procs]
        got = dict(q.get(timeout=120) for _ in range(2))
        [p.join(60) for p in procs]
        for r in range(2):
            assert torch.equal(got[r], want)


def test_single_rank_is_identity():
    x = torch.randn(3, 2)
    assert D.gather_clips(x, 3, 0, 1) is x
    assert D.clip_indices(5, 1, 2) == [1, 3]
