#!/usr/bin/env python3
"""bench.py -- scans/sec of the ICP registration hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic input: S independent scans of the
C2 workload (BASELINE.json configs[1]: ~120k-pt scan vs 1M-pt local map, 20 ICP iterations, fp32
points / fp64 accumulators), one context per scan, aligned together by mh_icp_align_batch in lock step (every
kernel of an iteration is one launch over all S scans), everything already resident in HBM when the timed
region starts.  With --gpus N (launched by torch.distributed.run) each
rank runs the same per-GPU batch on its own GPU (weak scaling, no data-path collective); the only
collective is the gather of the resulting poses (RCCL all_gather of 12 doubles per scan).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline":     algorithmic bytes of the match step of one scan / its HIP-event time per scan vs 8 TB/s HBM
  "cpu_baseline": the CPU oracle (a port of the reference algorithm, not the reference binary)
                  timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_query(p_bar: float) -> float:
    """12 (local xyz) + 27 x 16 (hash-slot probes) + 12 x P-bar (candidate points
    distance-tested) + 8 (pairing write-back)."""
    return 12.0 + 27.0 * 16.0 + 12.0 * p_bar + 8.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=32, help="scans per step and GPU (one context each; mh_icp_align_batch aligns them in lock step)")
    ap.add_argument("--workload", default="c2", choices=["c2", "creal", "small"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget (bounded sample)")
    ap.add_argument("--no-profile", action="store_true", help="do not time the match kernel with HIP events")
    args = ap.parse_args()

    import torch  # plumbing: process group, barrier, device selection
    import torch.distributed as dist
    from mola_lidar_odometry_amd import capi, synth
    from mola_lidar_odometry_amd import dist as mdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if not torch.cuda.is_available() or capi.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    w = {"c2": synth.workload_c2, "creal": synth.workload_creal, "small": synth.workload_small}[args.workload]()
    S = args.streams
    n_scan, n_map = len(w.scan_xyz), len(w.map_xyz)

    # ---- device-resident inputs (outside the timed region) --------------------------------------
    ctx0 = capi.Context(local_rank)
    gmap = capi.Map(ctx0, w.voxel_size, w.cap).build(w.map_xyz)
    ctxs = [capi.Context(local_rank) for _ in range(S)]
    tx = [torch.from_numpy(np.ascontiguousarray(w.scan_xyz[:, i])).cuda() for i in range(3)]
    scans = [capi.Scan.from_torch(c, *tx) for c in ctxs]
    rng = np.random.default_rng(1000 + rank)
    guesses = []
    for _ in range(S):  # C2 replicated: same scan, guesses jittered by 1 cm so the jobs are not byte-identical
        g = w.guess_ypr.copy()
        g[:3] += rng.normal(0, 0.01, 3)
        guesses.append(synth.pose_from_ypr(g))
    prof = not args.no_profile
    params = capi.ICPParams(max_iterations=w.n_iters, disable_stall_test=True, threshold=w.threshold,
                            kernel_param=w.kernel_param, poll_every=w.n_iters, profile=2 if prof else 0)
    maps = [gmap] * S

    def step():
        return capi.icp_align_batch(maps, scans, guesses, params)

    def sync_all():
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    match_ms, match_launches = 0.0, 0
    last = None
    for _ in range(args.steps):
        last = step()
        for r in last:
            match_ms += r["match_kernel_ms"]
            match_launches += r["n_match_launches"]
    sync_all()
    dt = time.perf_counter() - t0
    dt = mdist.max_over_ranks(dt, device="cuda" if distributed else None)  # MAX over ranks
    # outside the timed region: the same kernel with nothing else on the device (what a rocprofv3 kernel trace,
    # which serialises the streams, reports per launch)
    iso_ms, iso_launches = 0.0, 0
    if prof and rank == 0:
        for _ in range(3):
            for r in capi.icp_align_batch(maps[:1], scans[:1], guesses[:1], params):
                iso_ms += r["match_kernel_ms"]
                iso_launches += r["n_match_launches"]
    # the trivial result gather: poses of the last step from every rank, RCCL all_gather
    gathered = mdist.gather_poses(np.stack([r["T"] for r in last]), device="cuda" if distributed else None)
    all_poses = np.stack(gathered)

    scans_total = world * args.steps * S
    value = scans_total / dt

    out = {
        "metric": "scans/sec (KITTI 64-beam, ~120k pts, 1M-pt map) @1/2/4/8 GPU; ATE vs ref",
        "value": value, "unit": "scans/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 points / f64 accumulators", "data": "synthetic",
        "config": {"workload": f"{w.name}: {n_scan}-pt scan vs {n_map}-pt voxel-hashed map (voxel {w.voxel_size} m, "
                               f"cap {w.cap}), {w.n_iters} ICP iterations x 2 GN steps, GM kernel, sigma={w.sigma} "
                               "schedule of lidar3d-default.yaml:190,198",
                   "scans_per_step_per_gpu": S, "streams_per_gpu": S, "parallelism": f"{world}x independent GPUs"},
    }
    if rank == 0:
        # ---- CPU baseline: the oracle (a port), bounded sample, N=1 only ------------------------
        p_bar = None
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle_c
            om = oracle_c.Map(w.voxel_size, w.cap).insert(w.map_xyz)
            op = oracle_c.ICPParams(max_iterations=w.n_iters, disable_stall_test=True, threshold=w.threshold,
                                    kernel_param=w.kernel_param, compute_covariance=True)
            # pick the thread count that is fastest on THIS box (more threads than usable cores collapses
            # OpenMP throughput); the count actually used is what "cores" reports
            cores, best_t = min(8, oracle_c.max_threads()), None
            warm = oracle_c.ICPParams(max_iterations=1, disable_stall_test=True, threshold=w.threshold[:1],
                                      kernel_param=w.kernel_param[:1], compute_covariance=False)
            for nt in (8, 16, 24, 32, 48, 64, 96, 128):
                if nt > oracle_c.max_threads():
                    break
                oracle_c.icp_align(om, w.scan_xyz, guesses[0], warm, n_threads=nt)  # thread creation at this width
                tc = time.perf_counter()
                oracle_c.icp_align(om, w.scan_xyz, guesses[0], op, n_threads=nt)    # one FULL alignment, as sampled below
                tcal = time.perf_counter() - tc
                if best_t is None or tcal < best_t:
                    cores, best_t = nt, tcal
            n_done, t_cpu, o = 0, 0.0, None
            while t_cpu < args.cpu_seconds and n_done < 64:
                tc = time.perf_counter()
                o = oracle_c.icp_align(om, w.scan_xyz, guesses[0], op, n_threads=cores)
                t_cpu += time.perf_counter() - tc
                n_done += 1
            p_bar = o["n_candidates_total"] / (w.n_iters * n_scan)
            cpu = {"value": n_done / t_cpu, "unit": "scans/sec", "cores": cores, "kind": "port",
                   "sample": f"{n_done} full alignment(s) of the same workload ({w.n_iters} iterations each) "
                             f"with the C oracle (OpenMP, {cores} threads) in {t_cpu:.1f} s"}
            # parity of the timed product path against the oracle on the same input
            d = np.abs(last[0]["T"] - o["T"])
            out["parity_vs_cpu"] = {"max_abs_pose_diff": float(d.max()), "tolerance": 1e-4,
                                    "n_pairs_equal": bool(last[0]["n_final_pairs"] == o["n_final_pairs"])}
        if p_bar is None:
            stats = os.path.join(ROOT, "tests", "golden", "workload_stats.json")
            if os.path.exists(stats):
                p_bar = json.load(open(stats)).get(w.name, {}).get("p_bar")
        roof = None
        if prof and match_launches and p_bar:
            bytes_per_launch = n_scan * algorithmic_bytes_per_query(p_bar)
            avg_ms = match_ms / match_launches
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            kname = {"p": "k_match<fused,branch-and-bound>", "x": "k_match<fused,27-voxel>"}.get(
                os.environ.get("MH_MATCH", "q")[:1], "k_match4 (quad per point)")
            roof = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                    "avg_kernel_ms": avg_ms, "launches": match_launches,
                    "avg_kernel_ms_alone": (iso_ms / iso_launches) if iso_launches else None, "p_bar": p_bar,
                    "algorithmic_bytes_per_launch": bytes_per_launch,
                    "note": "avg_kernel_ms = the match step PER SCAN inside the timed region: HIP events around every lock-step match "
                            "launch (one launch over all scans of the step, blockIdx.y = scan) divided by the scans in it. The working set "
                            "(16 MB of records + the hash table) is cache resident: measured HBM traffic (`traffic`, bytes "
                            "per launch) is ~30x below the algorithmic bytes, so `achieved` is a rate of ALGORITHMIC bytes "
                            "and can exceed the HBM peak; what the kernel waits for is the chain of dependent L1-miss "
                            "round trips of its slowest wave (DESIGN.md section 3). avg_kernel_ms_alone = the same kernel "
                            "with a single stream, after the timed region: the figure a (stream-serialising) rocprofv3 "
                            "kernel trace reports"}
            import glob
            pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
            if pmc:  # HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes (profiles/collect.sh)
                summary = json.load(open(pmc[-1]))
                roof["traffic"] = summary.get("k_match_fused_hbm_bytes_per_launch")
                roof["traffic_source"] = os.path.basename(pmc[-1])
                # second view, since HBM is not what this kernel waits for: VALU issue.  SQ_INSTS_VALU wave-instructions per
                # scan / (1024 SIMD-32s x 2.4 GHz / 2 cycles per wave64 instruction, MI355X_MICROARCH.md) = the time the
                # match step of one scan needs if nothing but VALU issue limited it; its share of the measured time per
                # scan, alone and inside the lock-step batch
                mk = [k for k in summary.get("counters", {}) if k.startswith("k_match4")] if kname.startswith("k_match4") else []
                valu = summary["counters"][mk[0]].get("SQ_INSTS_VALU", {}).get("mean") if mk else None
                if valu and roof["avg_kernel_ms_alone"]:
                    floor_ms = valu / (1024 * 2.4e9 / 2.0) * 1e3
                    roof["valu"] = {"wave_instructions_per_launch": valu, "issue_floor_ms": floor_ms,
                                    "frac_of_launch_alone": floor_ms / roof["avg_kernel_ms_alone"],
                                    "frac_of_launch_concurrent": floor_ms / avg_ms}
        out["roofline"] = roof
        out["cpu_baseline"] = cpu
        out["gathered_poses"] = int(all_poses.shape[0] * all_poses.shape[1])
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
