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


def _runner_worker(rank, world, port, root, out_dir, q):
    """run_odometry.main under a world-size-2 gloo group with the GPU part (run_sequence) replaced by a stub."""
    import io
    import json
    from contextlib import redirect_stdout
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), MH_DIST_BACKEND="gloo")
    from mola_lidar_odometry_amd import run_odometry

    def fake_run_sequence(pipeline, scans, out_tum=None, device=None, prefetch=True):
        scans = list(scans)
        recs = [dict(timestamp=st, icp_good=True, map_updated=True, icp_iterations=3, n_for_icp=10, n_map_points=5) for st, _, _ in scans]
        traj = [(st, np.eye(4)[:3].reshape(12).tolist()) for st, _, _ in scans]
        open(out_tum, "w").write("".join("%f 0 0 0 0 0 0 1\n" % st for st, _, _ in scans))
        return recs, traj, 0.001 * len(scans) * (rank + 1)

    run_odometry.run_sequence = fake_run_sequence
    buf = io.StringIO()
    with redirect_stdout(buf):
        run_odometry.main(["--kitti-root", root, "--seqs", "00", "01", "02", "--out-dir", out_dir])
    q.put((rank, [json.loads(l) for l in buf.getvalue().strip().splitlines()]))


def test_sequence_runner_shards_whole_sequences_world2_gloo(tmp_path):
    """eval/cli_kitti.sh runs one sequence per worker; here: LPT assignment of whole sequences to ranks, one TUM per
    sequence, the summary on rank 0 counts every scan once and takes the slowest rank's time."""
    import torch.multiprocessing as mp
    root = tmp_path / "kitti"
    lengths = {"00": 5, "01": 2, "02": 4}
    for seq, n in lengths.items():
        d = root / "sequences" / seq / "velodyne"
        d.mkdir(parents=True)
        for k in range(n):
            np.zeros((7, 4), np.float32).tofile(d / ("%06d.bin" % k))
        np.savetxt(root / "sequences" / seq / "times.txt", 0.1 * np.arange(n))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_runner_worker, args=(r, 2, port, str(root), str(tmp_path / "out"), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seqs = {r: sorted(l["sequence"] for l in lines if "sequence" in l) for r, lines in res.items()}
    assert seqs == {0: ["00"], 1: ["01", "02"]}  # longest first: rank 0 gets 00 (5), rank 1 gets 02 (4) + 01 (2)
    summary = [l for l in res[0] if l.get("summary")]
    assert len(summary) == 1 and not any(l.get("summary") for l in res[1])
    assert summary[0]["scans"] == 11 and summary[0]["n_gpus"] == 2 and abs(summary[0]["seconds"] - 0.012) < 1e-9
    for seq, n in lengths.items():
        assert len(open(tmp_path / "out" / (seq + ".tum")).read().splitlines()) == n


def test_ranks_sharing_devices_plan():
    """run_odometry accepts more ranks than GPUs: device of each rank, GPUs in the job, and whether RCCL is out."""
    from mola_lidar_odometry_amd import dist as mdist
    assert mdist.plan_ranks_on_devices(8, 8, 8, 5) == (5, 8, False)           # one rank per GPU
    assert [mdist.plan_ranks_on_devices(4, 4, 1, r)[0] for r in range(4)] == [0, 0, 0, 0]
    assert mdist.plan_ranks_on_devices(4, 4, 1, 3) == (0, 1, True)            # four sequences on one GPU
    assert [mdist.plan_ranks_on_devices(16, 16, 8, r)[0] for r in (0, 7, 8, 15)] == [0, 7, 0, 7]
    assert mdist.plan_ranks_on_devices(16, 16, 8, 9) == (1, 8, True)
    assert mdist.plan_ranks_on_devices(32, 16, 8, 9) == (1, 16, True)         # two nodes
    assert mdist.plan_ranks_on_devices(2, 2, 0, 1) == (1, 2, False)           # CPU tests of the sharding logic (gloo)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without WORLD_SIZE starts N ranks itself (torch.distributed.run on 127.0.0.1) and the
    line it prints carries n_gpus == N (--launch-check: rendezvous and gather only, over gloo, no GPU)."""
    import json
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    got = json.loads(line)
    assert got["n_gpus"] == 2 and sorted(r[0] for r in got["ranks"]) == [0, 1]
