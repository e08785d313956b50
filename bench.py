#!/usr/bin/env python3
"""bench.py -- scans/sec of the ICP registration hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic input: S = 32 jobs (sequences), each with ITS OWN 1M-pt
local map (S independent draws of the generator: S sequences have S local maps, eval/cli_kitti.sh:23-36) and K = 4
DIFFERENT ~120k-pt sweeps against it (other sensor poses, other noise, own guesses -- consecutive scans of a sequence), i.e.
S x K = 128 scans of the C2 workload (BASELINE.json configs[1]: 20 ICP iterations, fp32 points / fp64 accumulators) per
step, aligned as K lock-step batches of S scans by mh_icp_align_batch, one context per job.  No two consecutive batches
see the same inputs.  As SURVEY.md 8(d) / BASELINE.md section 3 define the metric, the timed region contains, per scan:
the H->D copy of the scan (page-locked host memory, asynchronous, queued one batch ahead on the contexts of the other buffer
set so that it overlaps the current batch's kernels), the alignment, and the D->H copy of the result -- pose, covariance,
quality, counters AND Results::finalPairings (compacted on the device, downloaded on a copy stream that overlaps the next
batch).  The maps are built before the timed region (a local map changes only at key-frames).  `shared_map` repeats the
measurement with every job on ONE map and one scan (the best case for the caches, round 1's configuration) as a labelled
second number.

--gpus N: without WORLD_SIZE in the environment the script starts N ranks itself (torch.distributed.run on 127.0.0.1);
each rank runs the same per-GPU batch on its own GPU (weak scaling, no data-path collective), the only collective is
the gather of the resulting poses (RCCL all_gather of 12 doubles per scan).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline":     the match kernel (k_match_flat_b, one launch over all S scans of a batch): compulsory bytes per launch / its
                  HIP-event duration vs the 8 TB/s HBM peak (frac <= 1 by construction), the PMC-measured HBM traffic of
                  that kernel (profiles/) and the views that say what the kernel really waits for, each a counter of the
                  timed kernel over its clock cycles: vector-instruction issue with the measured mix, texture addresser /
                  data return, L1 tag rate and pending-line stalls, LDS, waves parked, and two sensitivity slopes;
  "cpu_baseline": the CPU oracle (a port of the reference algorithm, not the reference binary) timed on this box's
                  host cores on a bounded sample of the same workload (thread count used AND the box's logical cores);
and, at N = 1, labelled extras OUTSIDE `value` (--no-extras skips them):
  "single_sequence":     the config-3 proxy -- molahip-lo-cli (C++) on a 1000-sweep synthetic CITY drive (cross streets, houses,
                         cars, poles, trees; ~118 k raw points per sweep, skewed, per-point time stamps): whole-run and
                         steady scans/s, ICP-layer and local-map sizes, per-stage milliseconds, ATE against the generator's
                         ground truth, and the CPU oracle driver on the same scans (with its Python share taken out);
  "single_sequence_ndt": the config-5 proxy -- the same drive through pipelines/lidar3d-ndt-hip.yaml, with its CPU driver;
  "creal":               the 6 k-point layer, 32 in lock step, with its CPU figure;
  "multi_sequence":      N = 1, 2, 4, 8, 16 copies of the drive's first 400 scans through one molahip-lo-cli process.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L2_PEAK_GBPS = 34500.0  # ibid.: aggregate L2 bandwidth
VALU_WAVE_INSTR_PER_S = 1024 * 2.4e9 / 2.0  # 1024 SIMDs, one wave64 VALU instruction per two cycles at 2.4 GHz


def algorithmic_bytes_per_query(p_bar: float) -> float:
    """SURVEY.md 8(d), the REFERENCE algorithm's bytes per query point and iteration: 12 (local xyz) + 27 x 16
    (hash-slot probes) + 12 x P-bar (candidate points distance-tested) + 8 (pairing write-back)."""
    return 12.0 + 27.0 * 16.0 + 12.0 * p_bar + 8.0


def _gen(args):
    from mola_lidar_odometry_amd import synth
    if args[0] == "scanset":  # another sweep against workload (name, variant)'s map
        return synth.workload_scan_set(args[1], args[2], args[3])
    _, name, variant = args
    return synth.workload_by_name(name, variant)


def generate_inputs(name, variants, n_sets=1):
    """S independent draws of the generator and, per draw, n_sets - 1 further sweeps against the same map, in worker
    processes (numpy only; started before HIP is initialised).
    -> (workloads, sets): sets[j][k] = (scan_xyz, T_gt, T_guess) of job j's scan set k (set 0 = the workload's own)."""
    import multiprocessing as mp
    tasks = [("workload", name, v) for v in variants] + [("scanset", name, v, k) for v in variants for k in range(1, n_sets)]
    n_proc = max(1, min(len(tasks), (os.cpu_count() or 2) // 2, 32))
    if n_proc == 1:
        res = [_gen(t) for t in tasks]
    else:
        pool = mp.get_context("fork").Pool(n_proc)
        try:
            res = pool.map(_gen, tasks, chunksize=1)
        finally:
            # close + join, not terminate: under `rocprofv3 --pmc` a SIGTERMed worker enters the profiler's signal handler and
            # never returns (a whole collection run was lost to that)
            pool.close()
            pool.join()
    ws = res[:len(variants)]
    extra = res[len(variants):]
    sets = []
    for j, x in enumerate(ws):
        sets.append([(x.scan_xyz, x.T_gt, x.T_guess)] + [extra[j * (n_sets - 1) + k] for k in range(n_sets - 1)])
    return ws, sets


def generate_workloads(name, variants):
    return generate_inputs(name, variants)[0]


def compulsory_bytes(w, stats):
    """Bytes that have to move at least once per match launch and scan, whatever the search strategy: the scan (12 B per
    point), the pairing written back (16 B nearest point + d2, 4 B index), each map record / hash slot inside the union of
    the scan's 27-voxel neighbourhoods once (16 B each) and -- in every iteration but an alignment's first -- the previous
    pairing that bounds the search (16 B per point; averaged over the workload's iterations)."""
    n = len(w.scan_xyz)
    prev = 0.0 if os.environ.get("MH_NO_PREV_BOUND") or os.environ.get("MH_MATCH", "q")[:1] in "pxtw" else \
        16.0 * n * (w.n_iters - 1) / max(1, w.n_iters)
    return 12.0 * n + 20.0 * n + prev + 16.0 * stats["records_in_union"] + 16.0 * stats["voxels_in_union"]


def neighbourhood_union(w):
    """Occupied map voxels / stored points inside the union of the 27-voxel blocks around the scan points at the initial
    guess (numpy; bench bookkeeping for the compulsory-traffic figure, not the product path)."""
    T = w.T_guess.reshape(3, 4)
    p = (w.scan_xyz.astype(np.float64) @ T[:, :3].T + T[:, 3]).astype(np.float32)
    inv = np.float32(1.0) / np.float32(w.voxel_size)

    def keys(xyz):
        k = np.floor(xyz * inv).astype(np.int64) + (1 << 20)
        return (k[:, 0] << 42) | (k[:, 1] << 21) | k[:, 2]

    qk = np.unique(keys(p))
    nb = []
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                nb.append(qk + (dx << 42) + (dy << 21) + dz)
    nb = np.unique(np.concatenate(nb))
    mk, mc = np.unique(keys(w.map_xyz), return_counts=True)
    hit = np.isin(mk, nb)
    return {"query_voxels": int(len(qk)), "voxels_in_union": int(hit.sum()), "records_in_union": int(mc[hit].sum())}


CLI = os.path.join(ROOT, "mola_lidar_odometry_amd", "molahip-lo-cli")
PIPELINE = os.path.join(ROOT, "pipelines", "lidar3d-default-hip.yaml")
PIPELINE_NDT = os.path.join(ROOT, "pipelines", "lidar3d-ndt-hip.yaml")


def run_lo_cli(seq_dir, n_seq, out_stem, timeout=600, pipeline=PIPELINE, max_scans=None, scan_log=False, env=None):
    """molahip-lo-cli (C++, no Python in the loop) over n_seq copies of the sequence folder in ONE process: per-sequence
    reports, per-stage host milliseconds, the N-sequence summary line.  The synthetic drive carries per-point time stamps in
    the fourth float of a row (--time-field 12): the sweeps are skewed by the vehicle's motion, the de-skew filter has work."""
    cmd = [CLI, "--pipeline", pipeline, "--out", out_stem + ".tum", "--profile", "--time-field", "12"]
    if max_scans:
        cmd += ["--max-scans", str(max_scans)]
    if scan_log:
        cmd += ["--scan-log", "auto"]
    for _ in range(n_seq):
        cmd += ["--seq-dir", seq_dir]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **env) if env else None)
    if r.returncode != 0:
        raise RuntimeError("molahip-lo-cli failed (%d): %s" % (r.returncode, r.stderr[-800:]))
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    per = [l for l in lines if "sequence_dir" in l]
    prof = [l["profile_ms_per_scan"] for l in lines if "profile_ms_per_scan" in l]
    summary = next((l for l in lines if "sequences" in l), None)
    return per, prof, summary


def _city_drive_worker(root, n_scans, q):
    """(child process) cast the synthetic city drive into a KITTI-format folder; hands back the plan."""
    try:
        from mola_lidar_odometry_amd import synth_city
        t0 = time.perf_counter()
        seq_dir, drive = synth_city.write_kitti_drive(root, n_scans, time_channel=True)
        q.put(dict(seq_dir=seq_dir, poses=drive["poses"], stamps=drive["stamps"], points_per_scan=drive["points_per_scan"],
                   seconds=time.perf_counter() - t0))
    except BaseException as e:  # noqa: BLE001
        q.put(dict(error=repr(e)[:400]))


def start_city_drive(n_scans):
    """Start the generation of the extras' drive in a child process (OpenMP ray-caster, ~2 MB per sweep on disk) so that it
    runs beside the generation of the headline workload and the timed steps.  -> handle for finish_city_drive()."""
    import multiprocessing as mp
    import tempfile
    tmp = tempfile.TemporaryDirectory(prefix="molahip_bench_")
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    p = ctx.Process(target=_city_drive_worker, args=(tmp.name, n_scans, q), daemon=True)
    p.start()
    return dict(tmp=tmp, proc=p, queue=q)


def finish_city_drive(h, timeout=900):
    res = h["queue"].get(timeout=timeout)
    h["proc"].join(timeout=30)
    if "error" in res:
        raise RuntimeError("city drive generation failed: " + res["error"])
    return res


def cpu_driver_sample(pipeline, seq_dir, stamps, est, scan_seconds, cpu_seconds, label):
    """The CPU oracle driver (oracle/odometry_oracle.py: the per-scan control flow in Python, every point touched by the C
    oracle under OpenMP) on the first scans of the same folder, bounded by cpu_seconds.  Reports its rate over the scans it
    processed, the share of that time spent inside the C library (so that the interpreter's overhead is visible and can be
    taken out), and the device driver's rate over THE SAME scans."""
    from oracle import odometry_oracle as oo
    from oracle import oracle_c
    try:  # the driver's small numpy products must not wake a BLAS pool of one thread per logical core
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1, user_api="blas")
    except Exception:  # noqa: BLE001
        pass
    files = sorted(glob.glob(os.path.join(seq_dir, "velodyne", "*.bin")))

    def load(k):
        rows = np.fromfile(files[k], dtype=np.float32).reshape(-1, 4)
        return np.ascontiguousarray(rows[:, :3]), np.ascontiguousarray(rows[:, 3])

    # the thread count that is fastest on THIS box for THIS driver (a 1.6 k-point layer does not want 16 threads' worth
    # of fork/join): scans 0..29 per candidate, the C library's time over scans 10..29 decides
    threads, best_c, calib = min(16, oracle_c.max_threads()), None, {}
    for nt in (2, 4, 8, 16, 32):
        if nt > oracle_c.max_threads() or len(files) < 30:
            break
        oc_ = oo.OdometryOracle(pipeline, n_threads=nt)
        c_t = 0.0
        for k in range(30):
            xyz, t = load(k)
            oracle_c.reset_c_seconds()
            oc_.on_lidar(float(stamps[k] - stamps[0]), xyz, t)
            if k >= 10:
                c_t += oracle_c.C_SECONDS
        calib[str(nt)] = 20.0 / c_t if c_t > 0 else None
        if best_c is None or c_t < best_c:
            threads, best_c = nt, c_t
    o = oo.OdometryOracle(pipeline, n_threads=threads)
    t_steady, c_steady, k_steady, done, t_all = 0.0, 0.0, 0, 0, 0.0
    for k in range(len(files)):
        xyz, t = load(k)
        oracle_c.reset_c_seconds()
        tc = time.perf_counter()
        o.on_lidar(float(stamps[k] - stamps[0]), xyz, t)
        d = time.perf_counter() - tc
        t_all += d
        done += 1
        if k >= 5:
            t_steady += d
            c_steady += oracle_c.C_SECONDS
            k_steady += 1
        if t_all > cpu_seconds and k_steady >= 10:
            break
    rate = k_steady / t_steady if t_steady > 0 else None
    rate_c = k_steady / c_steady if c_steady > 0 else None
    out = {"value": rate, "unit": "scans/sec", "cores": threads, "host_logical_cores": os.cpu_count(), "kind": "port",
           "value_c_library_only": rate_c, "python_share_of_time": (1.0 - c_steady / t_steady) if t_steady > 0 else None,
           "thread_calibration_c_only_scans_per_s": calib,
           "sample": "%s: scans 5..%d of the same folder through the Python loop of oracle/odometry_oracle.py on the C oracle (OpenMP, %d "
                     "threads of %s logical cores), %.1f s; value_c_library_only counts only the time inside the C library (filters, "
                     "de-skew, matching, Gauss-Newton, covariance, map insertion)" % (label, done - 1, threads, os.cpu_count(), t_all)}
    if scan_seconds is not None and len(scan_seconds) >= done and done > 5:
        gpu_t = float(np.sum(scan_seconds[5:done]))
        out["device_driver_same_scans_per_s"] = (done - 5) / gpu_t if gpu_t > 0 else None
        if rate_c and gpu_t > 0:
            out["ratio_device_vs_cpu_c_only_same_scans"] = out["device_driver_same_scans_per_s"] / rate_c
            out["ratio_device_vs_cpu_with_python_same_scans"] = out["device_driver_same_scans_per_s"] / rate
        if calib.get("16"):  # the width SURVEY / BASELINE quote CPU figures at (16 threads), from the 20-scan calibration pass
            out["ratio_device_vs_cpu_c_only_16_threads_calibration_pass"] = out["device_driver_same_scans_per_s"] / calib["16"]
        # the ratio to QUOTE: against the best C-only figure the CPU showed anywhere in this run -- the timed
        # sample or any width of the 20-scan calibration pass
        best_cpu = max([v for v in [rate_c] + list(calib.values()) if v], default=None)
        out["best_cpu_c_only_scans_per_s_of_this_run"] = best_cpu
        if best_cpu and out.get("device_driver_same_scans_per_s"):
            out["ratio_device_vs_best_cpu_figure_of_this_run"] = out["device_driver_same_scans_per_s"] / best_cpu
    if est is not None and all("pose" in r for r in o.records):
        est_cpu = np.stack([r["pose"] for r in o.records]).reshape(-1, 3, 4)
        m = min(len(est), len(est_cpu))
        out["max_pose_diff_device_vs_cpu_m"] = float(np.abs(est[:m, :3, 3] - est_cpu[:m, :, 3]).max())
    return out


def _cpu_sequence_worker(args):
    """(child process) the CPU oracle driver over the first scans of the folder for `seconds`; -> (scans done after the
    first 5, seconds spent on them)."""
    pipeline, seq_dir, threads, seconds = args
    from oracle import odometry_oracle as oo
    from oracle import oracle_c
    files = sorted(glob.glob(os.path.join(seq_dir, "velodyne", "*.bin")))
    o = oo.OdometryOracle(pipeline, n_threads=threads)
    t_steady, c_steady, k_steady, t_all = 0.0, 0.0, 0, 0.0
    for k, f in enumerate(files):
        rows = np.fromfile(f, dtype=np.float32).reshape(-1, 4)
        oracle_c.reset_c_seconds()
        tc = time.perf_counter()
        o.on_lidar(0.1 * k, np.ascontiguousarray(rows[:, :3]), np.ascontiguousarray(rows[:, 3]))
        d = time.perf_counter() - tc
        t_all += d
        if k >= 5:
            t_steady += d
            c_steady += oracle_c.C_SECONDS  # (wall time inside the C library, the other processes competing for the cores included)
            k_steady += 1
        if t_all > seconds and k_steady >= 5:
            break
    return k_steady, t_steady, c_steady


def _cpu_c2_worker(args):
    """(child process) full C2 alignments with the C oracle (OpenMP, `threads` threads) for `seconds`; -> (alignments, seconds)."""
    variant, threads, seconds = args
    from mola_lidar_odometry_amd import synth
    from oracle import oracle_c
    w = synth.workload_by_name("c2", variant)
    om = oracle_c.Map(w.voxel_size, w.cap).insert(w.map_xyz)
    op = oracle_c.ICPParams(max_iterations=w.n_iters, disable_stall_test=True, threshold=w.threshold, kernel_param=w.kernel_param,
                            compute_covariance=True)
    warm = oracle_c.ICPParams(max_iterations=1, disable_stall_test=True, threshold=w.threshold[:1], kernel_param=w.kernel_param[:1],
                              compute_covariance=False)
    oracle_c.icp_align(om, w.scan_xyz, w.T_guess, warm, n_threads=threads)
    print("READY", flush=True)
    sys.stdin.readline()  # the parent releases all children together once every map is built
    n, t = 0, 0.0
    while t < seconds:
        tc = time.perf_counter()
        oracle_c.icp_align(om, w.scan_xyz, w.T_guess, op, n_threads=threads)
        t += time.perf_counter() - tc
        n += 1
    return n, t


def cpu_c2_throughput(n_proc, threads, seconds):
    """The headline workload on the CPU in THROUGHPUT mode: n_proc concurrent processes, each aligning its own
    C2 scan against its own 1M-point map with the C oracle on `threads` OpenMP threads -- as many as fill the box's logical cores --
    released together once every process has built its map.  Aggregate alignments per second."""
    env = dict(os.environ, OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", OMP_WAIT_POLICY="passive")
    procs = []
    for j in range(n_proc):
        code = ("import sys, json; sys.path.insert(0, %r); import bench; "
                "print(json.dumps(bench._cpu_c2_worker((%d, %d, %f))))" % (ROOT, j % 32, threads, seconds))
        procs.append(subprocess.Popen([sys.executable, "-c", code], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                      text=True, env=env))
    for pr in procs:  # wait until every child has generated its workload and built its map
        line = pr.stdout.readline()
        if not line.startswith("READY"):
            raise RuntimeError("a CPU worker did not start")
    for pr in procs:
        pr.stdin.write("go\n")
        pr.stdin.flush()
    res = []
    for pr in procs:
        out, _ = pr.communicate(timeout=300 + 10 * seconds)
        line = [l for l in out.splitlines() if l.startswith("[")]
        if pr.returncode == 0 and line:
            res.append(json.loads(line[-1]))
    if not res:
        raise RuntimeError("no CPU worker finished")
    return {"value": float(sum(n / t for n, t in res if t > 0)), "unit": "scans/sec", "processes": len(res), "threads_per_process": threads,
            "cores": len(res) * threads, "host_logical_cores": os.cpu_count(), "kind": "port",
            "sample": "%d concurrent processes x %d OpenMP threads, each full C2 alignments (%d total) of its own scan against its own "
                      "1M-point map with the C oracle for %.0f s" % (len(res), threads, sum(n for n, _ in res), seconds)}


def cpu_throughput(pipeline, seq_dir, n_proc, threads, seconds):
    """The CPU in THROUGHPUT mode (a one-at-a-time CPU figure against a batched GPU figure compares latency
    with throughput): n_proc copies of the CPU oracle driver at once, one process per sequence like eval/cli_kitti.sh:23 does
    with GNU parallel, `threads` OpenMP threads each.  Aggregate scans/s (Python loop included)."""
    code = ("import sys, json; sys.path.insert(0, %r); import bench; "
            "print(json.dumps(bench._cpu_sequence_worker((%r, %r, %d, %f))))" % (ROOT, pipeline, seq_dir, threads, seconds))
    # (numpy's BLAS pool would start one thread per logical core in EVERY child: 16 x 256 spinning threads starve each other)
    env = dict(os.environ, OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", OMP_WAIT_POLICY="passive")
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env)
             for _ in range(n_proc)]
    res = []
    for pr in procs:
        out, _ = pr.communicate(timeout=120 + 4 * seconds)
        line = [l for l in out.splitlines() if l.startswith("[")]
        if pr.returncode == 0 and line:
            res.append(json.loads(line[-1]))
    if not res:
        raise RuntimeError("no CPU driver process finished")
    return {"value": float(sum(r[0] / r[1] for r in res if r[1] > 0)), "unit": "scans/sec",
            "value_c_library_only": float(sum(r[0] / r[2] for r in res if r[2] > 0)),
            "processes": len(res), "threads_per_process": threads, "cores": len(res) * threads,
            "host_logical_cores": os.cpu_count(), "kind": "port",
            "sample": "%d concurrent CPU oracle drivers (one process per sequence, %d OpenMP threads each) on the first scans of the "
                      "drive, %.0f s each; value = with the Python loop of oracle/odometry_oracle.py, value_c_library_only = counting "
                      "only each process's time inside the C library (what a compiled driver would spend)" % (len(res), threads, seconds)}


def one_sequence(seq_dir, gt, stamps, n_raw, tmp, pipeline, tag, cpu_seconds, what):
    from mola_lidar_odometry_amd import synth_city, trajectory
    per, prof, _ = run_lo_cli(seq_dir, 1, os.path.join(tmp, tag), pipeline=pipeline, scan_log=True)
    p0 = per[0]
    _, est = trajectory.read_tum(p0["tum"])
    n = min(len(est), len(gt))
    path = synth_city.path_length(gt[:n, :3, :].reshape(n, 12))
    ate_o, ate_s = trajectory.ate_rmse(est[:n], gt[:n], "origin"), trajectory.ate_rmse(est[:n], gt[:n], "se3")
    scan_seconds = None
    log = p0["tum"][:-4] + "_scans.csv"
    if os.path.exists(log):
        scan_seconds = np.loadtxt(log, delimiter=",", skiprows=1, usecols=1)
    single = {"value": p0["steady_scans_per_s"], "unit": "scans/sec", "whole_run_scans_per_s": p0["scans_per_s"],
              "scans": p0["scans"], "good": p0["good"], "keyframes": p0["keyframes"],
              "icp_iterations_per_scan": p0["icp_iterations"] / max(1, p0["scans"]),
              "mean_raw_points": p0["mean_raw_points"], "mean_icp_layer_points": p0["mean_icp_points"],
              "mean_map_layer_points": p0["mean_map_layer_points"], "final_map_points": p0["final_map_points"],
              "max_map_points": p0["max_map_points"], "final_map_voxels": p0["final_map_voxels"],
              "ms_per_scan_by_stage": prof[0] if prof else None,
              "path_m": path, "ate_rmse_origin_m": ate_o, "ate_rmse_se3_m": ate_s, "ate_origin_pct_of_path": 100.0 * ate_o / path if path else None,
              "workload": "%d sweeps of ~%d raw points (HDL-64-like: 64 x 1875 rays, 80 m; synthetic city with cross streets, houses, parked "
                          "cars, poles, trees; the vehicle pulls away from rest, turns left and right; KITTI .bin rows with the point's time "
                          "stamp in the 4th float), %s: device filters + de-skew, ICP on the decimated layer, key-frame map updates; steady "
                          "state = registration time without the first 5 scans; file reading excluded; ATE against the generator's ground "
                          "truth" % (p0["scans"], int(np.mean(n_raw)), what),
              "driver": "molahip-lo-cli (C++), next-scan prefetch on"}
    try:
        single["cpu_driver"] = cpu_driver_sample(pipeline, seq_dir, stamps, est, scan_seconds, cpu_seconds, what)
        cd = single["cpu_driver"]
        # (against the BEST CPU figure of the run; `ratio_vs_cpu_driver_same_scans` is the timed sample's own, usually higher)
        single["ratio_vs_cpu_driver"] = cd.get("ratio_device_vs_best_cpu_figure_of_this_run") or cd.get("ratio_device_vs_cpu_c_only_same_scans")
        single["ratio_vs_cpu_driver_same_scans"] = cd.get("ratio_device_vs_cpu_c_only_same_scans")
    except Exception as e:  # noqa: BLE001  (an extra must not take the headline line down)
        single["cpu_driver"] = {"error": repr(e)[:300]}
    return single, p0


def sequence_extras(drive, tmp, seq_counts, cpu_seconds, log, multi_scans=400):
    """single_sequence / single_sequence_ndt / multi_sequence: the stand-alone odometry driver on the synthetic KITTI-format
    city drive -- the config-3 and config-5 proxies -- with the CPU oracle driver timed beside it on the same scans and the
    trajectory compared with the generator's ground truth; then N copies of the drive in one process.  Labelled extras,
    never part of `value`."""
    from mola_lidar_odometry_amd import synth_city
    out = {}
    seq_dir = drive["seq_dir"]
    gt = synth_city.ground_truth_44(drive)
    stamps = drive["stamps"]
    single, p0 = one_sequence(seq_dir, gt, stamps, drive["points_per_scan"], tmp, PIPELINE, "solo", cpu_seconds, "pipelines/lidar3d-default-hip.yaml")
    single["drive_generation_s"] = drive.get("seconds")
    out["single_sequence"] = single
    log("single_sequence done")
    try:
        ndt, _ = one_sequence(seq_dir, gt, stamps, drive["points_per_scan"], tmp, PIPELINE_NDT, "solo_ndt", cpu_seconds,
                              "pipelines/lidar3d-ndt-hip.yaml (mola::NDT map, Matcher_Point2Plane + Matcher_Points_DistanceThreshold)")
        out["single_sequence_ndt"] = ndt
    except Exception as e:  # noqa: BLE001
        out["single_sequence_ndt"] = {"error": repr(e)[:300]}
    log("single_sequence_ndt done")
    multi = {}
    identical = True
    solo_tum = None
    for c in [1] + [c for c in seq_counts if c > 1]:
        try:
            perc, _, summ = run_lo_cli(seq_dir, c, os.path.join(tmp, "multi%d" % c), max_scans=multi_scans)
            if c == 1:
                solo_tum = open(perc[0]["tum"]).read()
                # (whole run: the process's wall clock from before the HIP runtime is initialised to the last trajectory written, as for N > 1)
                multi["1"] = {"steady_scans_per_s": perc[0]["steady_scans_per_s"],
                              "whole_run_scans_per_s": summ["scans_per_s"] if summ else perc[0]["scans_per_s"]}
            else:
                multi[str(c)] = {"steady_scans_per_s": summ["steady_scans_per_s"], "whole_run_scans_per_s": summ["scans_per_s"]}
                identical = identical and all(open(q["tum"]).read() == solo_tum for q in perc)
        except Exception as e:  # noqa: BLE001
            multi[str(c)] = {"error": repr(e)[:300]}
    # the same question to the CPU: how many scans/s does the HOST deliver when it, too, runs many sequences at once?
    cpu_tp = None
    try:
        cores = os.cpu_count() or 8
        thr = int((single.get("cpu_driver") or {}).get("cores") or 8)
        n_proc = max(1, min(16, cores // max(1, thr)))
        cpu_tp = cpu_throughput(PIPELINE, seq_dir, n_proc, thr, min(6.0, cpu_seconds))
        best = max((v["steady_scans_per_s"] for v in multi.values() if "steady_scans_per_s" in v), default=None)
        if best and cpu_tp["value"]:
            cpu_tp["ratio_best_device_multi_sequence_vs_this_with_python"] = best / cpu_tp["value"]
            # the ratio to quote: against the C library's share only (the Python loop is the restatement's, not the reference's)
            cpu_tp["ratio_best_device_multi_sequence_vs_this"] = best / cpu_tp["value_c_library_only"]
            cpu_tp["ratio_by_sequence_count_vs_cpu_throughput_c_only"] = {
                k: v["steady_scans_per_s"] / cpu_tp["value_c_library_only"] for k, v in multi.items() if "steady_scans_per_s" in v}
    except Exception as e:  # noqa: BLE001
        cpu_tp = {"error": repr(e)[:300]}
    log("cpu throughput done")
    # ... and with a local map of ~1 M points (MOLA_LOCAL_MAP_MAX_SIZE, the reference pipeline's own knob, yaml:238): the default
    # radius max(100, 1.5 x range) keeps ~0.55-0.65 M points of this city
    try:
        perb, _, _ = run_lo_cli(seq_dir, 1, os.path.join(tmp, "bigmap"), env={"MOLA_LOCAL_MAP_MAX_SIZE": "250"})
        single["with_1M_point_local_map"] = {"steady_scans_per_s": perb[0]["steady_scans_per_s"], "whole_run_scans_per_s": perb[0]["scans_per_s"],
                                             "max_map_points": perb[0]["max_map_points"], "final_map_points": perb[0]["final_map_points"],
                                             "env": {"MOLA_LOCAL_MAP_MAX_SIZE": "250"}}
    except Exception as e:  # noqa: BLE001
        single["with_1M_point_local_map"] = {"error": repr(e)[:300]}
    out["multi_sequence"] = {"unit": "scans/sec", "sequences_in_one_process": multi, "trajectories_identical_to_solo_run": identical,
                             "scans_per_sequence": multi_scans, "cpu_throughput_mode": cpu_tp,
                             "note": "N copies of the first %d scans of the drive through ONE molahip-lo-cli process (a host thread per "
                                     "sequence, alignments merged into lock-step batches); steady = registration time of the slowest thread "
                                     "without its first 5 scans; whole run = wall clock incl. process start-up" % multi_scans}
    return out


def lpt_curve(seq_dir, tmp, devices, scale=20, timeout=900, time_field=True):
    """SURVEY 8(e)'s SECOND curve (config 4: the 11 KITTI sequences sharded one-sequence-per-GPU, eval/cli_kitti.sh:9,23-36), as one
    native command: eleven sequences with KITTI's length ratios (lengths / scale, the first scans of the synthetic city drive each)
    through `molahip-lo-cli --devices <list>` -- whole sequences onto the listed GPUs longest first -- with the makespan bound of that
    assignment beside the measured wall clock.  `devices`: "0" at N = 1, "0,1,..,N-1" on N GPUs (the same GPU may be listed several
    times: a batcher per slot, how the tests exercise the path on one GPU)."""
    from mola_lidar_odometry_amd import dist as mdist
    files = sorted(glob.glob(os.path.join(seq_dir, "velodyne", "*.bin")))
    stamps = open(os.path.join(seq_dir, "times.txt")).read().split() if os.path.exists(os.path.join(seq_dir, "times.txt")) else None
    lengths = [max(6, min(len(files), int(round(L / float(scale))))) for L in mdist.KITTI_SEQ_LENGTHS]
    root = os.path.join(tmp, "lpt")
    dirs = []
    for k, n in enumerate(lengths):
        d = os.path.join(root, "%02d" % k)
        os.makedirs(os.path.join(d, "velodyne"), exist_ok=True)
        for f in files[:n]:
            dst = os.path.join(d, "velodyne", os.path.basename(f))
            if not os.path.exists(dst):
                os.symlink(f, dst)
        if stamps:
            open(os.path.join(d, "times.txt"), "w").write("\n".join(stamps[:n]) + "\n")
        dirs.append(d)
    cmd = [CLI, "--pipeline", PIPELINE, "--out", os.path.join(root, "lpt.tum"), "--devices", devices]
    if time_field:  # the city drive carries per-point time stamps in the fourth float of a row
        cmd += ["--time-field", "12"]
    for d in dirs:
        cmd += ["--seq-dir", d]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    wall = time.perf_counter() - t0
    if r.returncode != 0:
        raise RuntimeError("molahip-lo-cli --devices failed (%d): %s" % (r.returncode, r.stderr[-600:]))
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    summ = next(l for l in lines if "sequences" in l)
    n_slots = len(devices.split(","))
    assign = mdist.lpt_assign(lengths, n_slots)
    mk = mdist.makespan(lengths, assign)
    return {"unit": "scans/sec", "value": summ["scans_per_s"], "steady_scans_per_s": summ.get("steady_scans_per_s"),
            "devices": devices, "device_slots": n_slots, "sequences": len(dirs), "sequence_scans": lengths, "scans": summ["scans"],
            "wall_seconds_cli": summ["wall_seconds"], "wall_seconds_incl_process_start": wall,
            "makespan_scans": mk, "speedup_bound_of_this_assignment": sum(lengths) / float(mk),
            "speedup_bound_kitti_8_gpus": sum(mdist.KITTI_SEQ_LENGTHS) / float(max(mdist.KITTI_SEQ_LENGTHS)),
            "per_device": summ.get("per_device"),
            "note": "whole sequences per GPU by longest-processing-time-first, no data-path collective (molahip-lo-cli --devices); the "
                    "speed-up over the one-device run of the same eleven sequences is bounded by total scans / makespan scans (4.98 for "
                    "KITTI 00-10 on 8 GPUs, SURVEY 8e); the driver divides this line's value at N GPUs by the N = 1 line's. "
                    "Unmeasured on more than one physical GPU until an 8-GPU node runs it."}


def relaunch_under_torchrun(n):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--streams", type=int, default=32, help="scans per step and GPU (one context each; mh_icp_align_batch aligns them in lock step)")
    ap.add_argument("--scan-sets", type=int, default=4,
                    help="batches per step: a step aligns this many DIFFERENT sweeps per job (other sensor poses, other noise, own guesses) "
                         "against the job's map, one lock-step batch each, so that consecutive batches never see identical inputs")
    ap.add_argument("--workload", default="c2", choices=["c2", "creal", "small"])
    ap.add_argument("--maps", default="distinct", choices=["distinct", "shared"],
                    help="distinct (default): every job has its own map and scan; shared: one map and scan for all jobs")
    ap.add_argument("--io", default="both", choices=["both", "upload", "pairs", "none"],
                    help="what travels inside the timed region: the scans H2D and the final pairings D2H (default, the metric's "
                         "definition), one of them, or nothing (resident inputs: an upper bound, not the metric)")
    ap.add_argument("--no-io", action="store_true", help="same as --io none")
    ap.add_argument("--upload-delay-ms", type=float, default=1.0, help="the upload thread starts this long after the step's batch call")
    ap.add_argument("--shared-stream", type=int, default=1, help="1: the contexts of a buffer set share one stream; 0: a stream per context")
    ap.add_argument("--upload-thread", type=int, default=1, help="1: the next step's uploads are queued by a second host thread while "
                    "the current step's batch call blocks; 0: by the main thread before the call")
    ap.add_argument("--no-shared-run", action="store_true", help="skip the second, shared-map measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget (bounded sample)")
    ap.add_argument("--no-profile", action="store_true", help="do not time the match kernel with HIP events")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the labelled extra measurements (single_sequence / creal / multi_sequence, N = 1 only, outside `value`)")
    ap.add_argument("--extras-scans", type=int, default=1000, help="length of the synthetic city drive of the sequence extras")
    ap.add_argument("--extras-cpu-seconds", type=float, default=8.0, help="budget of each CPU oracle driver sample of the extras")
    ap.add_argument("--extras-sequences", default="1,2,4,8,16", help="multi_sequence: sequences run together in one process")
    ap.add_argument("--no-lpt", action="store_true", help="skip the lpt_curve extra (11 sequences with KITTI's length ratios through molahip-lo-cli --devices)")
    ap.add_argument("--lpt-scale", type=float, default=20.0, help="lpt_curve: KITTI sequence lengths divided by this")
    ap.add_argument("--lpt-devices", default=None, help="lpt_curve: device list (default 0..N-1; e.g. 0,0,0,0 exercises four slots on one GPU)")
    ap.add_argument("--launch-check", action="store_true",
                    help="only start the ranks, gather their ranks over gloo and print n_gpus (no GPU needed: tests the self-launch)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch one rank per GPU")
    distributed = world > 1
    if args.launch_check:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        got = [None] * world
        dist.all_gather_object(got, (rank, local_rank))
        if rank == 0:
            print(json.dumps({"n_gpus": world, "ranks": got}), flush=True)
        dist.destroy_process_group()
        return
    S = args.streams
    n_var = 1 if args.maps == "shared" else S
    # inputs first (worker processes, before the HIP runtime exists in this one); ranks draw different variants
    want_extras = world == 1 and not args.no_extras and args.workload == "c2" and args.maps == "distinct"
    # the KITTI-length LPT curve is the one extra that also runs at N > 1: rank 0 starts molahip-lo-cli over all N devices
    want_lpt = not args.no_extras and not args.no_lpt and args.workload == "c2" and args.maps == "distinct"
    lpt_scans = int(round(max(__import__("mola_lidar_odometry_amd.dist", fromlist=["x"]).KITTI_SEQ_LENGTHS) / float(args.lpt_scale))) + 1
    city = start_city_drive(args.extras_scans if want_extras else lpt_scans) if (want_extras or want_lpt) and rank == 0 else None
    K = max(1, args.scan_sets)
    ws, sets = generate_inputs(args.workload, [rank * S + j for j in range(n_var)], K)
    w = ws[0]
    # the extras' drive is cast beside the workload generation above; it must be DONE before anything is timed (its 16
    # OpenMP threads and 1.9 GB of file writes next to a measurement cost the latency-bound extras a third of their rate)
    drive, drive_error = None, None
    if city is not None:
        try:
            drive = finish_city_drive(city)
        except Exception as e:  # noqa: BLE001
            drive_error = repr(e)[:300]

    import torch  # plumbing: pinned host memory, process group, barrier, device selection
    import torch.distributed as dist
    from mola_lidar_odometry_amd import capi, synth
    from mola_lidar_odometry_amd import dist as mdist

    if not torch.cuda.is_available() or capi.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    n_scan, n_map = len(w.scan_xyz), len(w.map_xyz)
    prof = not args.no_profile
    params = capi.ICPParams(max_iterations=w.n_iters, disable_stall_test=True, threshold=w.threshold,
                            kernel_param=w.kernel_param, poll_every=w.n_iters, profile=2 if prof else 0)

    class Setup:
        """Device-side state of one measurement: S jobs, two buffer sets (A/B) of contexts + scans, pinned host mirrors of the
        K scan sets every job rotates through (set k of a step is batch k of that step)."""

        def __init__(self, wl, scan_sets, io):
            self.wl, self.io = wl, io != "none"
            self.up, self.dl = io in ("both", "upload"), io in ("both", "pairs")
            self.thread, self.worker_error, self.calls, self.last_call = None, None, {}, None
            self.map_ctx = capi.Context(local_rank)
            self.maps = [capi.Map(self.map_ctx, x.voxel_size, x.cap).build(x.map_xyz) for x in wl]
            if len(self.maps) == 1:
                self.maps = self.maps * S
            self.job_w = [wl[j % len(wl)] for j in range(S)]
            self.job_sets = [scan_sets[j % len(wl)] for j in range(S)]
            self.K = len(self.job_sets[0]) if self.up else 1   # resident inputs: the scans never change, one set
            rng = np.random.default_rng(1000 + rank)
            self.guesses = []  # [set][job]
            for k in range(self.K):
                g_k = []
                for x, st in zip(self.job_w, self.job_sets):  # guesses jittered by 1 cm so that jobs sharing a scan are not byte-identical
                    T = np.array(st[k][2], np.float64).copy()
                    T[[3, 7, 11]] += rng.normal(0, 0.01, 3)
                    g_k.append(T)
                self.guesses.append(g_k)
            # page-locked host copy of every job's scans (what a driver would hand over): interleaved xyz records, the form a
            # sensor driver delivers, one copy per scan
            self.pinned = [[torch.from_numpy(np.ascontiguousarray(st[k][0], dtype=np.float32)).pin_memory() for st in self.job_sets]
                           for k in range(self.K)]
            self.sizes = [[len(st[k][0]) for st in self.job_sets] for k in range(self.K)]
            # one context per job and buffer set; the contexts of a set share ONE stream (the lock-step batch runs on its
            # first job's stream anyway): the uploads of the other set are then a second stream, not 32 -- HIP maps streams
            # onto a handful of hardware queues, and a copy + de-interleave pair waiting at the head of a queue it shares
            # with the batch's stream stalls the batch's kernels behind it (measured: +2 ms per step with 64 streams)
            n_bufs = 2 if self.up else 1
            self.streams = [torch.cuda.Stream(device=local_rank) for _ in range(n_bufs)] if args.shared_stream else None
            self.ctxs = [[capi.Context(local_rank, stream=self.streams[b].cuda_stream if self.streams else None) for _ in range(S)]
                         for b in range(n_bufs)]
            self.scans = [[capi.Scan(c, st[0][0]) for c, st in zip(cs, self.job_sets)] for cs in self.ctxs]
            nbytes = max(sum(capi.pairs_block_bytes(n) for n in sz) for sz in self.sizes)
            self.blocks = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(2)] if self.dl else None
            self.k = 0  # batches run so far: batch k reads buffer k & 1 and scan set k % K

        def upload(self, buf, sset, delay=0.0):
            if delay:  # (second host thread) let the main thread queue its batch first: both contend for the HIP runtime's locks
                time.sleep(delay)
            for sc, t, n in zip(self.scans[buf], self.pinned[sset], self.sizes[sset]):
                sc.update_interleaved_pinned(t.data_ptr(), n, 12)

        def _worker_main(self):  # the upload thread: one per Setup, fed through a queue (no thread start/join per batch)
            while True:
                item = self.todo.get()
                if item is None:
                    return
                try:
                    self.upload(item[0], item[1], args.upload_delay_ms * 1e-3)
                except BaseException as e:  # noqa: BLE001  (handed to the main thread)
                    self.worker_error = e
                self.done.put(item)

        def call_for(self, buf, cur, sset):
            """The marshalled arguments of a batch (capi.BatchCall): built once per (buffer, pairs block, scan set), reused every
            step -- the per-batch host work is the C call."""
            key = (buf, cur if self.dl else -1, sset)
            if key not in self.calls:
                if self.dl:
                    self.calls[key] = capi.BatchCall(self.maps, self.scans[buf], self.guesses[sset], params,
                                                     pairs_block=self.blocks[cur].data_ptr(), pairs_mem=capi.MEM_HOST_PINNED)
                else:
                    self.calls[key] = capi.BatchCall(self.maps, self.scans[buf], self.guesses[sset], params)
            return self.calls[key]

        def batch(self):
            """One lock-step batch; returns (match_kernel_ms, n_match_launches) of job 0 (lock step: its share of the launches)."""
            if not self.io:
                self.last_call = self.call_for(0, 0, 0)
                r = self.last_call.run()
                self.k += 1
                return r[0].match_kernel_ms, int(r[0].n_match_launches)
            cur = self.k & 1
            buf = cur if self.up else 0
            sset = self.k % self.K
            pending = False
            if self.up:  # next batch's scans: asynchronous copies on the OTHER buffer set's stream
                nxt = (1 - cur, (self.k + 1) % self.K)
                if args.upload_thread:  # queued by a second host thread while this one blocks in the batch call below
                    if self.thread is None:
                        import queue
                        import threading
                        self.todo, self.done = queue.SimpleQueue(), queue.SimpleQueue()
                        self.thread = threading.Thread(target=self._worker_main, daemon=True)
                        self.thread.start()
                    self.todo.put(nxt)
                    pending = True
                else:
                    self.upload(*nxt)
            self.last_call = self.call_for(buf, cur, sset)
            r = self.last_call.run()
            if pending:
                self.done.get()
                if self.worker_error is not None:
                    raise self.worker_error
            self.k += 1
            return r[0].match_kernel_ms, int(r[0].n_match_launches)

        def step(self):
            """One step = one pass over the job's K scan sets: K lock-step batches of S scans each."""
            ms, launches = 0.0, 0
            for _ in range(self.K):
                m, l = self.batch()
                ms += m
                launches += l
            return ms, launches

        def sync(self):
            for cs in self.ctxs:
                for c in cs:
                    c.synchronize()
            torch.cuda.synchronize()
            if distributed:
                dist.barrier()

        def run(self, steps, warmup):
            if self.up:
                self.upload(self.k & 1, self.k % self.K)
            for _ in range(warmup):
                self.step()
            self.sync()
            t0 = time.perf_counter()
            ms, launches = 0.0, 0
            for _ in range(steps):
                m, l = self.step()
                ms += m
                launches += l
            self.sync()
            dt = time.perf_counter() - t0
            dt = mdist.max_over_ranks(dt, device="cuda" if distributed else None)  # MAX over ranks
            return dt, ms, launches, self.last_call.results()

        def last_set(self):
            return (self.k - 1) % self.K

        def last_pairs(self, results):
            cur = (self.k - 1) & 1
            return capi.unpack_pairs_block(self.blocks[cur].numpy(), self.sizes[self.last_set()], results)

        def close(self):
            self.sync()
            if self.thread is not None:
                self.todo.put(None)
                self.thread.join(timeout=10)
                self.thread = None
            self.calls.clear()
            for cs in self.ctxs:
                for c in cs:
                    c.close()
            self.map_ctx.close()

    io = "none" if args.no_io else args.io
    main_run = Setup(ws, sets, io)
    KS = main_run.K  # batches (= scan sets) per step
    dt, match_ms, match_launches, last = main_run.run(args.steps, args.warmup)
    # outside the timed region: the same kernel with nothing else on the device (S = 1), what rocprofv3 --stats of a
    # one-stream run reports per launch
    iso_ms, iso_launches = 0.0, 0
    if prof and rank == 0:
        for _ in range(3):
            for r in capi.icp_align_batch(main_run.maps[:1], main_run.scans[0][:1], main_run.guesses[0][:1], params):
                iso_ms += r["match_kernel_ms"]
                iso_launches += r["n_match_launches"]
    # the trivial result gather: poses of the last step from every rank, RCCL all_gather
    gathered = mdist.gather_poses(np.stack([r["T"] for r in last]), device="cuda" if distributed else None)
    all_poses = np.stack(gathered)
    pairs_last = main_run.last_pairs(last) if main_run.dl else None
    if pairs_last is not None:
        pairs_last = [dict(local_idx=p["local_idx"].copy(), global_idx=p["global_idx"].copy(), d2=p["d2"].copy()) for p in pairs_last]
    last_set = main_run.last_set()
    guesses = main_run.guesses[last_set]     # the inputs of the LAST timed batch: what `last` / `pairs_last` belong to
    last_scans = [st[last_set][0] for st in main_run.job_sets]
    guess0 = main_run.guesses[0][0]
    main_run.close()

    shared = None
    if args.maps == "distinct" and not args.no_shared_run and S > 1:
        sh = Setup(ws[:1], [sets[0][:1]], io)
        sdt, sms, sl, _ = sh.run(max(2, args.steps // 2), 1)
        shared = {"value": world * max(2, args.steps // 2) * S / sdt, "unit": "scans/sec",
                  "match_kernel_ms_per_scan": (sms / sl) if sl else None,
                  "note": "every job on ONE map and ONE scan (guesses jittered): the working set of all jobs is 20 MB, the "
                          "best case for L1/L2/Infinity Cache; round 1's configuration"}
        sh.close()

    # ---- labelled extras (N = 1 only, outside `value`, bounded): VERDICT r2 item 2 ---------------------------------
    extras = {}
    t_extras = time.perf_counter()

    def elog(msg):
        print("[bench extras %.1f s] %s" % (time.perf_counter() - t_extras, msg), file=sys.stderr, flush=True)

    if want_extras and rank == 0:
        # creal: what lidar3d-default.yaml really feeds align() -- a ~6 k-point layer (C-real) -- against the SAME
        # 32 maps, in lock step, I/O in the timed region like the headline
        try:
            import dataclasses
            cws = []
            for j, x in enumerate(ws):
                v = rank * S + j
                scene = synth.make_scene(12345 + 1009 * v, 120.0, 40)
                cws.append(dataclasses.replace(x, name="Creal_6k_vs_1M" + ("#%d" % v if v else ""),
                                               scan_xyz=synth.make_scan(scene, x.pose_gt_ypr, 32, 192, 54321 + 31 * v)))
            cr = Setup(cws, [[(x.scan_xyz, x.T_gt, x.T_guess)] for x in cws], io)
            csteps = max(10, args.steps)
            cdt, cms, cl, clast = cr.run(csteps, 3)
            creal = {"value": csteps * S / cdt, "unit": "scans/sec", "ms_per_step": 1e3 * cdt / csteps, "scans_per_step": S,
                     "points_per_scan": int(np.mean([len(x.scan_xyz) for x in cws])),
                     "match_kernel_ms_per_scan": (cms / cl) if cl else None,
                     "workload": "32 x 192 ray-cast scan (~6 k points) vs the same 1M-pt maps, 20 ICP iterations, lock step, "
                                 "scan H2D + result D2H incl. finalPairings inside the timed region"}
            cguess = cr.guesses[0]
            cr.close()
            from oracle import oracle_c
            opc = oracle_c.ICPParams(max_iterations=cws[0].n_iters, disable_stall_test=True, threshold=cws[0].threshold,
                                     kernel_param=cws[0].kernel_param, compute_covariance=True)
            omc = oracle_c.Map(cws[0].voxel_size, cws[0].cap).insert(cws[0].map_xyz)
            best = None
            for nt in (4, 8, 16, 24):
                if nt > oracle_c.max_threads():
                    break
                oracle_c.icp_align(omc, cws[0].scan_xyz, cguess[0], opc, n_threads=nt)
                tc = time.perf_counter()
                k = 0
                while time.perf_counter() - tc < 0.5:
                    oc_res = oracle_c.icp_align(omc, cws[0].scan_xyz, cguess[0], opc, n_threads=nt)
                    k += 1
                rate = k / (time.perf_counter() - tc)
                if best is None or rate > best[0]:
                    best = (rate, nt)
            creal["cpu_baseline"] = {"value": best[0], "unit": "scans/sec", "cores": best[1], "kind": "port",
                                     "sample": "job 0's alignment repeated for 0.5 s per thread count with the C oracle"}
            creal["max_abs_pose_diff_vs_cpu_job0"] = float(np.abs(clast[0]["T"] - oc_res["T"]).max())
            extras["creal"] = creal
        except Exception as e:  # noqa: BLE001  (an extra must not take the headline line down)
            extras["creal"] = {"error": repr(e)[:300]}
        elog("creal done")
        try:
            if drive is None:
                raise RuntimeError(drive_error or "no drive")
            extras.update(sequence_extras(drive, city["tmp"].name, [int(v) for v in args.extras_sequences.split(",") if v],
                                          args.extras_cpu_seconds, elog))
        except Exception as e:  # noqa: BLE001
            extras.setdefault("single_sequence", {"error": repr(e)[:300]})
        elog("sequence extras done")
    if want_lpt and rank == 0:
        try:
            if drive is None:
                raise RuntimeError(drive_error or "no drive")
            devs = args.lpt_devices or ",".join(str(d) for d in range(world))
            extras["lpt_curve"] = lpt_curve(drive["seq_dir"], city["tmp"].name, devs, args.lpt_scale)
        except Exception as e:  # noqa: BLE001
            extras["lpt_curve"] = {"error": repr(e)[:300]}
        elog("lpt_curve done")
    if city is not None:
        city["tmp"].cleanup()

    scans_total = world * args.steps * S * KS
    value = scans_total / dt

    out = {
        "metric": "scans/sec (KITTI 64-beam, ~120k pts, 1M-pt map) @1/2/4/8 GPU; ATE vs ref",
        "value": value, "unit": "scans/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 points / f64 accumulators", "data": "synthetic",
        "config": {"workload": f"{ws[0].name}: {n_scan}-pt scan vs {n_map}-pt voxel-hashed map (voxel {w.voxel_size} m, "
                               f"cap {w.cap}), {w.n_iters} ICP iterations x 2 GN steps, GM kernel, sigma={w.sigma} "
                               "schedule of lidar3d-default.yaml:190,198",
                   "scans_per_step_per_gpu": S * KS, "scans_per_batch": S, "batches_per_step": KS,
                   "inputs_per_step": "every job aligns %d different sweeps per step (other sensor poses along the street, other noise, "
                                      "own guesses) against its map, one lock-step batch of %d scans each" % (KS, S),
                   "maps": args.maps,
                   "distinct_maps_per_gpu": len(ws), "map_bytes_per_gpu": int(len(ws) * (n_map * 16 + 4 * 2 ** 20)),
                   "timed_region": {"both": "scan H2D (pinned, async, one step ahead) + align + result D2H incl. finalPairings",
                                    "upload": "scan H2D (pinned, async, one step ahead) + align + result D2H without finalPairings",
                                    "pairs": "align (scans resident) + result D2H incl. finalPairings",
                                    "none": "align only, inputs resident (--io none)"}[io],
                   "parallelism": f"{world}x independent GPUs"},
        "shared_map": shared,
        "real_data": {"KITTI_BASE_DIR": os.environ.get("KITTI_BASE_DIR"), "MULRAN_BASE_DIR": os.environ.get("MULRAN_BASE_DIR"),
                      "available": bool(os.environ.get("KITTI_BASE_DIR") and os.path.isdir(os.environ.get("KITTI_BASE_DIR", "")))},
    }
    if rank == 0:
        # ---- CPU baseline: the oracle (a port), bounded sample, N=1 only ------------------------
        p_bar = None
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle_c
            op = oracle_c.ICPParams(max_iterations=w.n_iters, disable_stall_test=True, threshold=w.threshold,
                                    kernel_param=w.kernel_param, compute_covariance=True)
            om = oracle_c.Map(w.voxel_size, w.cap).insert(w.map_xyz)
            # pick the thread count that is fastest on THIS box (more threads than usable cores collapses
            # OpenMP throughput); the count actually used is what "cores" reports
            cores, best_t = min(8, oracle_c.max_threads()), None
            warm = oracle_c.ICPParams(max_iterations=1, disable_stall_test=True, threshold=w.threshold[:1],
                                      kernel_param=w.kernel_param[:1], compute_covariance=False)
            for nt in (8, 16, 24, 32, 48, 64, 96, 128):
                if nt > oracle_c.max_threads():
                    break
                oracle_c.icp_align(om, w.scan_xyz, guess0, warm, n_threads=nt)  # thread creation at this width
                tcal = None
                for _ in range(2):  # (the better of two FULL alignments, as sampled below: one alone picked a poor width now and then)
                    tc = time.perf_counter()
                    oracle_c.icp_align(om, w.scan_xyz, guess0, op, n_threads=nt)
                    d = time.perf_counter() - tc
                    tcal = d if tcal is None or d < tcal else tcal
                if best_t is None or tcal < best_t:
                    cores, best_t = nt, tcal
            n_done, t_cpu, o = 0, 0.0, None
            while t_cpu < args.cpu_seconds and n_done < 64:
                tc = time.perf_counter()
                o = oracle_c.icp_align(om, w.scan_xyz, guess0, op, n_threads=cores)
                t_cpu += time.perf_counter() - tc
                n_done += 1
            p_bar = o["n_candidates_total"] / (w.n_iters * n_scan)
            cpu = {"value": n_done / t_cpu, "unit": "scans/sec", "cores": cores, "host_logical_cores": os.cpu_count(), "kind": "port",
                   "mode": "latency: one alignment at a time on the fastest thread count of this box",
                   "sample": f"{n_done} full alignment(s) of job 0's workload ({w.n_iters} iterations each) "
                             f"with the C oracle (OpenMP, {cores} threads of {os.cpu_count()} logical cores) in {t_cpu:.1f} s"}
            # ... and in THROUGHPUT mode, as the device figure is one: as many concurrent alignments as fill the box's logical cores
            try:
                n_proc = max(1, min(32, (os.cpu_count() or cores) // cores))
                cpu["throughput_mode"] = cpu_c2_throughput(n_proc, cores, min(10.0, args.cpu_seconds))
            except Exception as e:  # noqa: BLE001
                cpu["throughput_mode"] = {"error": repr(e)[:300]}
            # parity of the timed product path against the oracle: EVERY job of the last timed step
            worst, pairs_equal, idx_equal = 0.0, True, True
            for j in range(S):
                x = ws[j % len(ws)]
                omj = om if j % len(ws) == 0 else oracle_c.Map(x.voxel_size, x.cap).insert(x.map_xyz)
                oj = oracle_c.icp_align(omj, last_scans[j], guesses[j], op, n_threads=cores, want_pairs=pairs_last is not None)
                worst = max(worst, float(np.abs(last[j]["T"] - oj["T"]).max()))
                pairs_equal = pairs_equal and last[j]["n_final_pairs"] == oj["n_final_pairs"]
                if pairs_last is not None:
                    pj = oj["pairs"]
                    idx_equal = idx_equal and np.array_equal(pairs_last[j]["local_idx"], pj["local_idx"]) and \
                        np.array_equal(pairs_last[j]["global_idx"], pj["global_idx"]) and np.array_equal(pairs_last[j]["d2"], pj["d2"])
            out["parity_vs_cpu"] = {"jobs_checked": S, "scan_set_checked": last_set, "max_abs_pose_diff": worst, "tolerance": 1e-4,
                                    "n_pairs_equal": bool(pairs_equal),
                                    "downloaded_final_pairings_bit_equal": bool(idx_equal) if pairs_last is not None else None}
        stats_file = os.path.join(ROOT, "tests", "golden", "workload_stats.json")
        if p_bar is None and os.path.exists(stats_file):
            p_bar = json.load(open(stats_file)).get(w.name.split("#")[0], {}).get("p_bar")
        roof = None
        if prof and match_launches:
            union = neighbourhood_union(w)
            per_scan = compulsory_bytes(w, union)
            avg_ms = match_ms / match_launches          # per scan: launch duration / scans in the launch
            launch_ms = avg_ms * S
            achieved = per_scan * S / (launch_ms * 1e-3) / 1e9
            kname = {"p": "k_match<fused,branch-and-bound>", "x": "k_match<fused,27-voxel>", "t": "k_match_tile_b",
                     "w": "k_match_wave_dense_b + k_match_wave_sparse_b", "o": "k_match4o_b",
                     "q": "k_match4_b (quad per point, one launch over all scans of a batch)"}.get(
                os.environ.get("MH_MATCH", "f")[:1], "k_match_flat_b (plan / scan: a wave per 64 points, one launch over all scans of a batch)")
            roof = {"bound": "valu+texture-path",
                    "bound_note": "what the counters say limits this kernel (views: vector-instruction issue, texture addresser / data "
                                  "return, L1 tag rate, LDS, waves parked -- none saturated alone); `achieved` / `frac` stay on COMPULSORY "
                                  "bytes against the HBM peak, the contract's hbm-style fraction",
                    "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                    "avg_launch_ms": launch_ms, "scans_per_launch": S, "launches": match_launches,
                    "avg_kernel_ms_per_scan": avg_ms,
                    "avg_kernel_ms_alone": (iso_ms / iso_launches) if iso_launches else None,
                    "compulsory_bytes_per_launch": per_scan * S, "compulsory": union,
                    "note": "achieved = COMPULSORY bytes per launch (scan read + previous pairing read + pairing write + every map "
                            "record / hash slot inside the union of the scan's 27-voxel blocks once) / HIP-event duration of the launch, so frac "
                            "<= 1 is the share of the HBM peak this launch would need if nothing was cached; `traffic` = "
                            "PMC-measured HBM bytes per launch of the same kernel (profiles/).  The kernel is NOT HBM-bound: "
                            "see `views`."}
            views = {}
            if p_bar:
                ref_bytes = n_scan * algorithmic_bytes_per_query(p_bar)
                views["reference_algorithm_bytes"] = {
                    "p_bar": p_bar, "bytes_per_scan_launch": ref_bytes, "rate_GBps": ref_bytes / (avg_ms * 1e-3) / 1e9,
                    "note": "SURVEY 8(d) bytes of the REFERENCE's exhaustive 27-voxel scan per second of this kernel; the "
                            "branch-and-bound search does not move them, so this is a speed-up statement, not a roofline"}
            pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
            if pmc:
                summary = json.load(open(pmc[-1]))
                roof["traffic_source"] = os.path.basename(pmc[-1])
                tr = summary.get("timed_kernel", {})
                if tr.get("hbm_bytes_per_launch") is not None:
                    roof["traffic"] = tr["hbm_bytes_per_launch"]
                    views["hbm_measured"] = {"bytes_per_launch": tr["hbm_bytes_per_launch"], "scans_per_launch": tr.get("scans_per_launch"),
                                             "frac_of_peak": tr["hbm_bytes_per_launch"] / (tr.get("avg_launch_ms", launch_ms) * 1e-3) / 1e9 / HBM_PEAK_GBPS
                                             if tr.get("avg_launch_ms") else None}
                if tr.get("l2_request_bytes_per_launch") is not None and tr.get("avg_launch_ms"):
                    views["l2"] = {"request_bytes_per_launch": tr["l2_request_bytes_per_launch"],
                                   "frac_of_peak": tr["l2_request_bytes_per_launch"] / (tr["avg_launch_ms"] * 1e-3) / 1e9 / L2_PEAK_GBPS}
                # Ceilings that are ceilings, every one from counters of the TIMED kernel in `traffic_source`
                # (separate PMC passes of this command); shares of the launch's clock cycles (GRBM_GUI_ACTIVE / 8 XCDs)
                cyc = tr.get("launch_cycles_under_pmc")
                if tr.get("valu_wave_instructions_per_launch") is not None and tr.get("avg_launch_ms"):
                    n_valu = tr["valu_wave_instructions_per_launch"]
                    mix = tr.get("valu_mix") or {}
                    slow = sum(mix.get(k, 0.0) for k in ("ADD_F64", "MUL_F64", "INT64"))  # issued over 4 cycles, the rest over 2
                    floor_ms = n_valu / VALU_WAVE_INSTR_PER_S * 1e3
                    views["valu"] = {"wave_instructions_per_launch": n_valu, "mix_per_launch": mix,
                                     "issue_floor_ms_at_2_cycles": floor_ms, "frac_of_launch": floor_ms / tr["avg_launch_ms"],
                                     "frac_of_launch_cycles_with_mix": ((n_valu - slow) * 2.0 + slow * 4.0) / 1024.0 / cyc if cyc else None,
                                     "lanes_active_per_instruction": (tr["SQ_THREAD_CYCLES_VALU"] / n_valu) if tr.get("SQ_THREAD_CYCLES_VALU") else None,
                                     "note": "vector instructions x 2 cycles (fp64 / 64-bit integer: 4) / 1024 SIMDs over the launch's cycles"}
                if cyc:
                    if tr.get("TCP_TOTAL_CACHE_ACCESSES_sum") is not None:
                        views["l1_tag_rate"] = {"tag_lookups_per_launch": tr["TCP_TOTAL_CACHE_ACCESSES_sum"],
                                                "frac_of_one_lookup_per_cycle_and_cu": tr["TCP_TOTAL_CACHE_ACCESSES_sum"] / (256.0 * cyc),
                                                "pending_stall_frac": (tr["TCP_PENDING_STALL_CYCLES_sum"] / (256.0 * cyc)) if tr.get("TCP_PENDING_STALL_CYCLES_sum") else None,
                                                "l1_hit_rate": tr.get("l1_hit_rate"),
                                                "note": "TCP_TOTAL_CACHE_ACCESSES / (256 CUs x cycles); pending_stall: cycles a request waited for a line already on its way"}
                    if tr.get("ta_busy_frac") is not None:
                        views["texture_path"] = {"ta_busy_frac": tr["ta_busy_frac"], "td_busy_frac": tr.get("td_busy_frac"),
                                                 "vector_memory_reads_per_launch": tr.get("SQ_INSTS_VMEM_RD"),
                                                 "floor_frac_at_17_cycles_per_read": (tr["SQ_INSTS_VMEM_RD"] * 17.0 / 256.0 / cyc) if tr.get("SQ_INSTS_VMEM_RD") else None,
                                                 "note": "TA_TA_BUSY / TD_TD_BUSY over 256 CUs x cycles; a dwordx4 read that hits in L1 costs the data "
                                                         "return >= 16-17 cycles per wave-instruction and CU, a miss ~2.3 cycles per 128-byte line "
                                                         "(tools/l1_cost.hip, profiles/r05_match_kernel.md section 3)"}
                    if tr.get("SQ_LDS_IDX_ACTIVE") is not None:
                        views["lds"] = {"busy_frac": tr["SQ_LDS_IDX_ACTIVE"] / 256.0 / cyc, "instructions_per_launch": tr.get("SQ_INSTS_LDS"),
                                        "atomics_per_launch": tr.get("SQ_INSTS_LDS_ATOMIC")}
                if tr.get("wait_frac") is not None:
                    views["wave_wait_frac"] = tr["wait_frac"]
            sens = os.path.join(ROOT, "profiles", "r05_sensitivity.json")
            if os.path.exists(sens) and os.environ.get("MH_MATCH", "f")[:1] == "f":
                sj = json.load(open(sens))
                views["sensitivity"] = {"source": "profiles/r05_sensitivity.json", "vector_memory_slope": sj["loads_twice"]["slope"],
                                        "valu_slope": sj["valu_plus"]["slope"], "note": sj["reading"] + " (relative slow-down per relative "
                                        "amount of ADDED work of that kind; what-if runs that remove work change the search and are useless here)"}
            roof["views"] = views
        out["roofline"] = roof
        out["cpu_baseline"] = cpu
        out.update(extras)  # single_sequence / creal / multi_sequence: labelled, outside `value`
        out["gathered_poses"] = int(all_poses.shape[0] * all_poses.shape[1])
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
