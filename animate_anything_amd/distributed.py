"""Batch-of-clips sharding across the GPUs of one node.

The reference's inference is single-process (train.py:825-857).  Clips are independent units - each
denoising loop only touches its own latents/prompt/mask (models/pipeline.py:163-198) - so the N-GPU
path is: one process per GPU, rank r takes clips r, r+W, ..., weights replicated, NO collective in the
data path, and ONE all-gather of the final latents (0.5 MB per clip at 16x64x64 fp16) over RCCL/xGMI.
"""
from __future__ import annotations

import os
from typing import List, Sequence

import torch
import torch.distributed as dist


def init(backend: str | None = None):
    """Initialise torch.distributed from the torchrun environment (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*).
    backend None -> "nccl" (= RCCL on ROCm) when a GPU is visible, else "gloo".  Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    cuda = torch.cuda.is_available()
    device = torch.device("cuda", local) if cuda else torch.device("cpu")
    if cuda:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if cuda else "gloo")
        if backend == "nccl" and cuda:
            dist.init_process_group(backend, device_id=device)      # binds the RCCL communicator to this rank's GPU
        else:
            dist.init_process_group(backend)
    return rank, world, device


def pin_to_gpu_numa_node(device_index: int):
    """Best effort: restrict this process to the CPUs of the NUMA node its GPU hangs off (eight ranks on a two-socket host
    otherwise run their Python launch loops wherever the scheduler puts them).  Returns the CPU set used, or None when the
    topology cannot be read (containers without /sys, single-node hosts: nothing to do)."""
    try:
        props = torch.cuda.get_device_properties(device_index)
        bdf = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = set()
            for part in f.read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
            return cpus
    except (OSError, ValueError, AttributeError, RuntimeError):
        pass
    return None


def clip_indices(num_clips: int, rank: int, world: int) -> List[int]:
    """Round-robin ownership: rank r denoises clips r, r+world, ..."""
    return list(range(rank, num_clips, world))


def clip_seed(base_seed: int, clip_index: int) -> int:
    """Per-clip seed so results do not depend on the number of ranks."""
    return base_seed + clip_index


def gather_clips(local: torch.Tensor, num_clips: int, rank: int, world: int) -> torch.Tensor:
    """All-gather per-rank results [n_local, ...] into [num_clips, ...] in clip order on every rank.
    One collective; ranks with fewer clips pad to the common count."""
    if world == 1:
        return local
    per = (num_clips + world - 1) // world
    shape = (per,) + tuple(local.shape[1:])
    padded = torch.zeros(shape, dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty((world,) + shape, dtype=local.dtype, device=local.device)
    if dist.get_backend() == "gloo":
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded)
        out = torch.stack(parts)
    else:
        dist.all_gather_into_tensor(out, padded)
    # out[r, j] is clip r + j*world
    order = out.transpose(0, 1).reshape((per * world,) + tuple(local.shape[1:]))
    return order[:num_clips].contiguous()


# ----------------------------------------------------------------------------------------- guidance-parallel (latency) mode
_pair_groups = {}


def guidance_pair(rank: int, world: int):
    """Optional single-clip latency mode: the two halves of a classifier-free-guidance batch (same
    latents, unconditional / conditional context) run on the two GPUs of a pair and exchange their noise predictions every step
    (0.5 MB per rank at 16x64x64) instead of running back to back on one GPU.  Ranks (2p, 2p+1) form pair p; returns
    (pair index, role, process group) with role 0 = unconditional half, 1 = conditional half.  Every rank must call this (the
    sub-groups are created collectively)."""
    if world % 2:
        raise ValueError("guidance-parallel mode needs an even number of ranks")
    if world not in _pair_groups:
        _pair_groups[world] = [dist.new_group([2 * p, 2 * p + 1]) for p in range(world // 2)]
    return rank // 2, rank % 2, _pair_groups[world][rank // 2]


def all_gather_cat(t: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate `t` of every rank of `group` along dim 0, in rank order (one all-gather: RCCL on the GPUs, gloo in the CPU tests)."""
    n = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo":
        parts = [torch.empty_like(t) for _ in range(n)]
        dist.all_gather(parts, t.contiguous(), group=group)
        return torch.cat(parts)
    out = torch.empty((n * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    return out
