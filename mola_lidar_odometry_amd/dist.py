"""Multi-GPU plumbing for the hot path.

The path shards only ACROSS trajectories/scans: within one sequence scan k+1 needs scan k's pose
(LidarOdometry.cpp:810-811, 1031-1039), so one process per GPU takes whole sequences (what the
reference's eval/cli_kitti.sh:23-36 does with GNU parallel) or, for synthetic batches, an equal share
of independent scans.  There is no data-path collective; the only exchange is the final gather of the
estimated poses (a few KB), done with torch.distributed (backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

# KITTI odometry sequence lengths 00..10: the config-4 sharding problem
KITTI_SEQ_LENGTHS = [4541, 1101, 4661, 801, 271, 2761, 1101, 1101, 4071, 1591, 1201]


def lpt_assign(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of whole sequences to ranks.  Returns, per rank, the
    list of sequence ids (deterministic: ties broken by id)."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    loads = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += lengths[i]
    return out


def makespan(lengths: Sequence[int], assignment: List[List[int]]) -> int:
    return max(sum(lengths[i] for i in seqs) if seqs else 0 for seqs in assignment)


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced share of n_items independent scans for `rank` (strong-scaling batches)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def gather_poses(local_poses: np.ndarray, device=None) -> List[np.ndarray]:
    """All-gather ragged [n_i, 12] float64 pose arrays from every rank (the 'trivial result gather').
    Works on any initialised torch.distributed backend; returns the per-rank arrays in rank order."""
    import torch
    import torch.distributed as dist

    local = np.ascontiguousarray(local_poses, dtype=np.float64).reshape(-1, 12)
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    world = dist.get_world_size()
    dev = device if device is not None else "cpu"
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts) if counts else 0
    buf = torch.zeros((max(nmax, 1), 12), dtype=torch.float64, device=dev)
    if local.shape[0]:
        buf[: local.shape[0]] = torch.from_numpy(local).to(dev)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    return [b[:c].cpu().numpy() for b, c in zip(bufs, counts)]


def max_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def plan_ranks_on_devices(world: int, local_world: int, n_devices: int, local_rank: int):
    """How the ranks of one node use its GPUs -> (device index of this rank, number of distinct GPUs in the job,
    oversubscribed?).  One rank per GPU is the normal case; with more local ranks than GPUs (several sequences per GPU,
    each in its own process -- they overlap well) the ranks share the devices round robin, and
    the caller must not use RCCL (one rank per device) for its reductions."""
    if n_devices <= 0 or local_world <= n_devices:
        return local_rank, world, False
    nodes = max(1, world // max(local_world, 1))
    return local_rank % n_devices, n_devices * nodes, True
