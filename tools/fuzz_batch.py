"""development aid: random batches (2-10 jobs; layers of 300-60 k points = every lock-step kernel chain and the per-stream
fallback; point maps and an NDT map; per-job budgets, schedules, solver settings, priors, hooks, stall test, final pairings) through mh_icp_align_batch against the same
jobs run one by one: results bitwise equal."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import capi, synth  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
ctx0 = capi.Context(0)
scene = synth.make_scene(4242, 70.0, 20)
maps = [capi.Map(ctx0, 1.0, 20).build(synth.make_map(scene, 120000, 4242)),
        capi.Map(ctx0, 0.5, 8).build(synth.make_map(scene, 60000, 4243)),
        capi.Map(ctx0, 2.0, 12, 0, 0.1, 0.05, 4).build(synth.make_map(scene, 120000, 4244))]  # an NDT map: Matcher_Point2Plane rides along
NDT = 2
pose = [1.0, -0.5, synth.SENSOR_H, 0.04, 0.003, -0.002]
cloud = synth.make_scan(scene, pose, rings=64, azimuths=1000, seed=9)
bad = 0
for case in range(n_cases):
    n_jobs = int(rng.integers(2, 11))
    ctxs = [capi.Context(0) for _ in range(n_jobs)]
    sizes = [int(rng.choice([300, 900, 2000, 3000, 6000, 12000, 20000, 40000, 60000])) for _ in range(n_jobs)]
    if rng.integers(0, 3) == 0:  # a uniform batch: one lock-step group
        sizes = [sizes[0]] * n_jobs
    scans = [capi.Scan(c, cloud[rng.permutation(len(cloud))[:n]]) for c, n in zip(ctxs, sizes)]
    jm = [maps[int(rng.choice([0, 0, 1, NDT]))] for _ in range(n_jobs)]
    guesses, ps, priors = [], [], []
    for j in range(n_jobs):
        g = np.array(pose) + np.concatenate([rng.normal(0, 0.15, 3) * [1, 1, 0.1], rng.normal(0, 0.01, 3)])
        guesses.append(synth.pose_from_ypr(g))
        iters = int(rng.choice([3, 8, 25, 60]))
        thr, kp = synth.threshold_schedule(float(rng.choice([1.0, 2.0])), iters)
        extra = {}
        if jm[j] is maps[NDT]:
            extra["pt2pl_threshold"] = float(rng.choice([0.4, 0.8]))
        if rng.integers(0, 4) == 0:
            extra.update(hook_enabled=True, hook_min_trans=float(rng.choice([0.05, 0.2])), hook_min_rot=float(np.deg2rad(0.75)))
        gn = capi.GNParams(max_inner_iterations=int(rng.choice([1, 2, 2, 4])), robust_kernel=int(rng.integers(0, 6)),
                           min_delta=float(rng.choice([0.0, 1e-7, 1e-4])), max_cost=float(rng.choice([0.0, 0.0, 1e-3])))
        ps.append(capi.ICPParams(max_iterations=iters, threshold=thr, kernel_param=kp, disable_stall_test=bool(rng.integers(0, 2)),
                                 poll_every=int(rng.choice([0, 4, iters])), gn=gn, **extra))
        priors.append((guesses[-1], np.diag([4.0, 4.0, 4.0, 100.0, 100.0, 100.0])) if rng.integers(0, 4) == 0 else None)
    singles = []
    for j, (m, s, g, p, pr) in enumerate(zip(jm, scans, guesses, ps, priors)):
        try:
            singles.append(capi.icp_align(m, s, g, p, prior=pr, want_trace=False, want_pairs=True))
        except capi.MolahipError as e:
            print("single job failed:", e, "n=%d iters=%d stall_off=%s poll=%d prior=%s map=%d" % (
                sizes[j], p.max_iterations, p.disable_stall_test, p.poll_every, pr is not None, maps.index(m)), flush=True)
            raise
    block = np.zeros(sum(capi.pairs_block_bytes(n) for n in sizes), np.uint8)
    batch = capi.icp_align_batch(jm, scans, guesses, ps, priors=priors if any(priors) else None, pairs_block=block)
    ok = True
    for a, b, pr in zip(singles, batch, capi.unpack_pairs_block(block, sizes, batch)):
        ok = ok and (a["n_iterations"], a["termination_reason"], a["n_final_pairs"]) == (b["n_iterations"], b["termination_reason"], b["n_final_pairs"])
        ok = ok and np.array_equal(a["T"], b["T"]) and np.array_equal(a["cov"], b["cov"]) and a["quality"] == b["quality"]
        ok = ok and all(np.array_equal(pr[k], a["pairs"][k]) for k in ("local_idx", "global_idx", "global_xyz", "d2"))
    bad += 0 if ok else 1
    print("case %2d jobs=%2d sizes=%s -> %s" % (case, n_jobs, sizes, "ok" if ok else "MISMATCH"), flush=True)
    for s in scans:
        s.close()
    for c in ctxs:
        c.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
