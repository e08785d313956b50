"""N>1 path on CPU: world_size-2 gloo process group exercising the sharding and the result gather that
bench.py uses with RCCL on the GPU box (shard across sequences/scans, no data-path
collective, one trivial gather of poses)."""
import os
import socket

import numpy as np
import pytest

from mola_lidar_odometry_amd import dist as mdist


def test_lpt_assignment_properties():
    L = mdist.KITTI_SEQ_LENGTHS
    assert sum(L) == 23201
    for world in (1, 2, 4, 8):
        a = mdist.lpt_assign(L, world)
        assert sorted(i for r in a for i in r) == list(range(11))  # every sequence exactly once
        assert mdist.makespan(L, a) >= max(max(L), -(-sum(L) // world))
    # at 8 GPUs the makespan is the longest sequence (02: 4661) -> ceiling 23201/4661 = 4.98x
    assert mdist.makespan(L, mdist.lpt_assign(L, 8)) == 4661
    assert mdist.makespan(L, mdist.lpt_assign(L, 1)) == 23201
    assert mdist.lpt_assign(L, 2) == mdist.lpt_assign(L, 2)  # deterministic


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 100, 1001):
        for world in (1, 2, 3, 8):
            parts = [mdist.shard_range(n, r, world) for r in range(world)]
            flat = [i for p in parts for i in p]
            assert flat == list(range(n))
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # each rank "aligned" a different number of scans (ragged): poses tagged with rank and index
        n_local = 3 + 2 * rank
        local = np.zeros((n_local, 12))
        local[:, 0] = rank
        local[:, 3] = np.arange(n_local)
        parts = mdist.gather_poses(local)
        t = mdist.max_over_ranks(1.0 + rank)
        # sequences sharded one-per-rank (LPT) and the empty-share edge case
        mine = mdist.lpt_assign(mdist.KITTI_SEQ_LENGTHS, world)[rank]
        empty = mdist.gather_poses(np.zeros((0, 12)) if rank == 1 else np.ones((2, 12)))
        q.put((rank, [p.tolist() for p in parts], t, mine, [e.shape[0] for e in empty]))
    finally:
        dist.destroy_process_group()


def test_gather_of_poses_world2_gloo():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, parts0, t0, mine0, e0), (r1, parts1, t1, mine1, e1) = res
    assert parts0 == parts1  # every rank sees the same gathered result, in rank order
    assert [len(p) for p in parts0] == [3, 5]
    for rank, p in enumerate(parts0):
        p = np.asarray(p)
        assert np.all(p[:, 0] == rank) and p[:, 3].tolist() == list(range(len(p)))
    assert t0 == t1 == 2.0  # max over ranks (the bench's timing rule)
    assert sorted(mine0 + mine1) == list(range(11)) and not set(mine0) & set(mine1)
    assert e0 == e1 == [2, 0]
