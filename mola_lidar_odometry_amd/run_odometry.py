"""Run the stand-alone odometry driver over sequences and time it (BASELINE.json configs[2]/[3]).

    python -m mola_lidar_odometry_amd.run_odometry --synthetic 200                       # synthetic drive, 200 scans
    python -m mola_lidar_odometry_amd.run_odometry --kitti-root /data/kitti --seqs 00 04  # KITTI velodyne folders
    python -m torch.distributed.run --nproc-per-node 8 ... run_odometry.py --kitti-root ... --seqs 00 ... 10
    (more ranks than GPUs is allowed and pays off: the ranks share the devices round robin, each sequence in its own process)

What eval/cli_kitti.sh:23-50 (relative to /root/reference) does with mola-lidar-odometry-cli + GNU parallel: one whole
sequence per worker (here one process per GPU, sequences assigned longest-first), a TUM trajectory per sequence,
then the KITTI / ATE metrics when ground truth (poses/XX.txt) is present.  Prints one JSON line per sequence and a
summary line on rank 0.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import time

import numpy as np

from . import dist as mdist
from . import synth, trajectory

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _kitti_scans(seq_dir):
    files = sorted(glob.glob(os.path.join(seq_dir, "velodyne", "*.bin")))
    times = os.path.join(seq_dir, "times.txt")
    stamps = np.loadtxt(times) if os.path.exists(times) else 0.1 * np.arange(len(files))
    for f, st in zip(files, stamps):
        rec = np.fromfile(f, dtype=np.float32).reshape(-1, 4)  # x,y,z,intensity rows: uploaded as they are
        yield float(st), rec, None  # KITTI odometry scans are motion compensated and carry no time stamps


def _mulran_scans(seq_dir):
    """A MulRan sequence folder (eval/cli_mulran.sh:23-36; apps/mola-lidar-odometry-cli.cpp:186-208 `--input-mulran-seq`):
    <dir>/sensor_data/Ouster/<time stamp in ns>.bin (or <dir>/Ouster/), float32 x,y,z,intensity rows."""
    d = os.path.join(seq_dir, "sensor_data", "Ouster")
    if not os.path.isdir(d):
        d = os.path.join(seq_dir, "Ouster")
    files = sorted(glob.glob(os.path.join(d, "*.bin")), key=lambda f: int(os.path.splitext(os.path.basename(f))[0]))
    t0 = int(os.path.splitext(os.path.basename(files[0]))[0]) if files else 0
    for f in files:
        rec = np.fromfile(f, dtype=np.float32).reshape(-1, 4)
        yield 1e-9 * (int(os.path.splitext(os.path.basename(f))[0]) - t0), rec, None


def is_mulran_dir(seq_dir):
    return not os.path.isdir(os.path.join(seq_dir, "velodyne")) and (
        os.path.isdir(os.path.join(seq_dir, "sensor_data", "Ouster")) or os.path.isdir(os.path.join(seq_dir, "Ouster")))


def sequence_scans(seq_dir):
    """KITTI or MulRan folder -> iterator of (stamp, records [n,4], None)."""
    return _mulran_scans(seq_dir) if is_mulran_dir(seq_dir) else _kitti_scans(seq_dir)


def mulran_gt(seq_dir):
    """global_pose.csv of a MulRan sequence: rows `stamp_ns, r00, r01, r02, tx, r10, ... tz` -> (stamps [s], poses [n,4,4])
    or None.  (The vehicle's pose; the Ouster sits 1.7 m ahead and 1.8 m up, turned by ~180 deg: compare after an SE(3)
    fit -- what `evo_ape -a` does, eval/cli_mulran.sh:50 -- and mind that a lever arm remains.)"""
    f = os.path.join(seq_dir, "global_pose.csv")
    if not os.path.exists(f):
        return None
    a = np.loadtxt(f, delimiter=",").reshape(-1, 13)
    T = np.tile(np.eye(4), (len(a), 1, 1))
    T[:, :3, :] = a[:, 1:].reshape(-1, 3, 4)
    return 1e-9 * a[:, 0], T


def _kitti_gt(root, seq, Tr=None):
    f = os.path.join(root, "poses", seq + ".txt")
    if not os.path.exists(f):
        return None
    a = np.loadtxt(f).reshape(-1, 3, 4)
    T = np.tile(np.eye(4), (len(a), 1, 1))
    T[:, :3] = a
    if Tr is not None:  # camera-frame ground truth -> velodyne frame
        T = np.linalg.inv(Tr)[None] @ T @ Tr[None]
    return T


def _kitti_calib_Tr(seq_dir):
    f = os.path.join(seq_dir, "calib.txt")
    if not os.path.exists(f):
        return None
    for line in open(f):
        if line.startswith("Tr:"):
            return np.vstack([np.asarray(line.split()[1:], np.float64).reshape(3, 4), [0, 0, 0, 1]])
    return None


def run_sequence(pipeline, scans, out_tum=None, device=None, prefetch=True):
    """scans: iterable of (stamp, xyz[N,3] fp32, t[N] fp32 | None).  Returns (records, trajectory, seconds).
    prefetch: announce scan k+1 before registering scan k (its upload + first filter pass overlap with scan k's ICP)."""
    from . import _mp2p_icp_hip as H
    from . import capi
    capi.lib()  # torch's HIP runtime first (one runtime per process), then libmolahip
    if device is None:
        device = int(os.environ.get("MH_DEVICE", os.environ.get("LOCAL_RANK", 0)))
    lo = H.LidarOdometry(device=device, own_context=True)
    lo.initialize(H.Config.FromYamlFile(pipeline))
    t0 = time.perf_counter()
    n = 0
    per_scan = []
    def as_f32(item):  # the very arrays handed to prefetch must be the ones handed to onLidar (matched by address)
        st, xyz, t = item
        return (st, np.ascontiguousarray(xyz, dtype=np.float32), None if t is None else np.ascontiguousarray(t, dtype=np.float32))

    it = iter(scans)
    cur = next(it, None)
    cur = as_f32(cur) if cur is not None else None
    while cur is not None:
        nxt = next(it, None)
        nxt = as_f32(nxt) if nxt is not None else None
        t1 = time.perf_counter()
        if prefetch and nxt is not None:
            lo.prefetch(nxt[1], nxt[2])
        lo.onLidar(cur[0], cur[1], cur[2])
        per_scan.append(time.perf_counter() - t1)
        n += 1
        cur = nxt
    dt = time.perf_counter() - t0
    # the first scans pay for the device context, the code objects and the first map: quote the steady state apart
    if out_tum:
        lo.saveTrajectoryTUM(out_tum)
    recs = lo.records()
    if recs:  # per-run figures travel with the records (several runs may be in flight in threads of this process)
        recs[-1]["_steady_scans_per_s"] = (len(per_scan) - 3) / sum(per_scan[3:]) if len(per_scan) > 3 and sum(per_scan[3:]) > 0 else 0.0
        recs[-1]["_startup_s"] = sum(per_scan[:3])
        recs[-1]["_host_ms_per_scan"] = {k: round(1e3 * v / max(n, 1), 4) for k, v in lo.profile().items()}
    return recs, lo.trajectory(), dt


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--pipeline", default=os.path.join(ROOT, "pipelines", "lidar3d-default-hip.yaml"))
    ap.add_argument("--synthetic", type=int, default=0, help="number of scans of the synthetic drive")
    ap.add_argument("--rings", type=int, default=64)
    ap.add_argument("--azimuths", type=int, default=1875, help="64 x 1875 = the 120k-point sweep of BASELINE.json C2")
    ap.add_argument("--kitti-root", default=None, help="KITTI odometry root (sequences/XX/velodyne, poses/XX.txt)")
    ap.add_argument("--mulran-root", default=None, help="MulRan root (<seq>/sensor_data/Ouster/*.bin, <seq>/global_pose.csv): "
                    "what MULRAN_BASE_DIR is to eval/cli_mulran.sh; --seqs then names KAIST01 DCC02 ...")
    ap.add_argument("--seqs", nargs="*", default=[])
    ap.add_argument("--copies", type=int, default=1, help="run the synthetic drive this many times (as separate sequences)")
    ap.add_argument("--no-prefetch", action="store_true", help="strictly sequential scans (what a live sensor feed gives)")
    ap.add_argument("--out-dir", default="gpurun_out/odometry")
    a = ap.parse_args(argv)

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    backend = os.environ.get("MH_DIST_BACKEND", "nccl")  # "gloo" in the CPU tests of the sharding logic
    n_gpus = world
    if world > 1:
        import torch
        import torch.distributed as td
        ndev = torch.cuda.device_count() if backend == "nccl" else 0
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        dev, n_gpus, shared = mdist.plan_ranks_on_devices(world, local_world, ndev, int(os.environ.get("LOCAL_RANK", 0)))
        if backend == "nccl" and shared:
            # more ranks than GPUs (several sequences per GPU, each in its own process): the only collective is a count
            # and a max, and RCCL wants one rank per device -> gloo
            backend = "gloo"
            os.environ["MH_DEVICE"] = str(dev)
        elif backend == "nccl":
            torch.cuda.set_device(dev)
        td.init_process_group(backend)
    red_dev = "cuda" if backend == "nccl" else "cpu"
    os.makedirs(a.out_dir, exist_ok=True)

    jobs = []  # (name, length, factory)
    drive_cache = {}
    for c in range(a.copies if a.synthetic else 0):
        jobs.append(("synthetic" if a.copies == 1 else "synthetic%d" % c, a.synthetic, None))
    for s in a.seqs:
        if a.mulran_root:
            d = os.path.join(a.mulran_root, s)
            n_files = len(glob.glob(os.path.join(d, "sensor_data", "Ouster", "*.bin"))) or len(glob.glob(os.path.join(d, "Ouster", "*.bin")))
            jobs.append((s, n_files, d))
            continue
        d = os.path.join(a.kitti_root, "sequences", s)
        jobs.append((s, len(glob.glob(os.path.join(d, "velodyne", "*.bin"))), d))
    mine = mdist.lpt_assign([j[1] for j in jobs], world)[rank]

    def do_job(j):
        name, length, src = jobs[j]
        if src is None:
            drive = drive_cache["drive"]
            scans = [(st, xyz, t) for (xyz, t), st in zip(drive["scans"], drive["stamps"])]
            G = np.stack([trajectory.to44(p) for p in drive["poses"]])
            gt = np.linalg.inv(G[0])[None] @ G
            gt_stamps = drive["stamps"]
        elif is_mulran_dir(src):
            scans = _mulran_scans(src)
            g = mulran_gt(src)
            gt, gt_stamps = (None, None)
            if g is not None:  # ground-truth stamps on the scans' relative clock
                files = sorted(glob.glob(os.path.join(src, "sensor_data", "Ouster", "*.bin")) or glob.glob(os.path.join(src, "Ouster", "*.bin")),
                               key=lambda f: int(os.path.splitext(os.path.basename(f))[0]))
                t0 = 1e-9 * int(os.path.splitext(os.path.basename(files[0]))[0]) if files else 0.0
                gt_stamps, gt = g[0] - t0, g[1]
        else:
            scans = _kitti_scans(src)
            gt = _kitti_gt(a.kitti_root, name, _kitti_calib_Tr(src))
            gt_stamps = None
        out = os.path.join(a.out_dir, "%s.tum" % name)
        recs, traj, secs = run_sequence(a.pipeline, scans, out, prefetch=not a.no_prefetch)
        line = dict(sequence=name, scans=len(recs), seconds=secs, scans_per_s=len(recs) / secs if secs else 0.0,
                    good=int(sum(r["icp_good"] for r in recs)), keyframes=int(sum(r["map_updated"] for r in recs)),
                    icp_iterations=int(sum(r["icp_iterations"] for r in recs)),
                    mean_points_for_icp=float(np.mean([r["n_for_icp"] for r in recs])) if recs else 0.0,
                    map_points=int(recs[-1]["n_map_points"]) if recs else 0, tum=out, rank=rank,
                    steady_scans_per_s=recs[-1].get("_steady_scans_per_s", 0.0) if recs else 0.0,
                    startup_s_first_3_scans=recs[-1].get("_startup_s", 0.0) if recs else 0.0,
                    host_ms_per_scan=recs[-1].get("_host_ms_per_scan", {}) if recs else {})
        if gt is not None and len(traj):
            est_stamps = np.array([t for t, _ in traj])
            est = np.stack([trajectory.to44(p) for _, p in traj])
            if gt_stamps is None:
                gt_stamps = np.array([r["timestamp"] for r in recs])[: len(gt)]
            ia, ib = trajectory.associate(est_stamps, np.asarray(gt_stamps))
            if len(ia) > 1:
                g = np.linalg.inv(gt[ib[0]])[None] @ gt[ib]
                e = np.linalg.inv(est[ia[0]])[None] @ est[ia]
                line["ate_rmse_m"] = trajectory.ate_rmse(e, g, "none")
                te, re, k = trajectory.kitti_relative_errors(e, g)
                if k:
                    line["kitti_t_err_percent"], line["kitti_r_err_deg_per_m"] = te, re
        print(json.dumps(line), flush=True)
        return len(recs), secs

    if any(jobs[j][2] is None for j in mine):  # one synthetic drive, shared by its copies
        drive_cache["drive"] = synth.make_drive(a.synthetic, rings=a.rings, azimuths=a.azimuths)
    # (Sequences side by side on one GPU pay off only as separate PROCESSES -- launch more ranks than GPUs, see below:
    #  4 ranks on one MI355X register 1617 scans/s together against 737 for one; threads of one process sharing the HIP
    #  runtime measured slower than one after the other, 483 vs 722 scans/s.)
    done = [do_job(j) for j in mine]
    total_time = sum(d[1] for d in done)  # the time spent registering scans
    total_scans = sum(d[0] for d in done)
    wall = mdist.max_over_ranks(total_time, device=red_dev if world > 1 else None)
    if world > 1:
        import torch
        import torch.distributed as td
        tot = torch.tensor([total_scans], dtype=torch.int64, device=red_dev)
        td.all_reduce(tot)
        total_scans = int(tot.item())
        td.destroy_process_group()
    if rank == 0:
        print(json.dumps(dict(summary=True, n_gpus=n_gpus, ranks=world, scans=total_scans, seconds=wall,
                              scans_per_s=total_scans / wall if wall else 0.0)), flush=True)


if __name__ == "__main__":
    main()
