"""development aid: random alignments (voxel size, cap, index mode, guess error, layer size, matcher) against the CPU oracle --
per-iteration pair counts, final pairings and d2 bit for bit, poses to 1e-9.  Exercises the previous-pairing bound and its
fallback (large guess errors with small voxels) beyond the fixed cases of tests/test_gpu_parity.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import capi, synth  # noqa: E402
from oracle import oracle_c  # noqa: E402

oracle_c.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
ctx = capi.Context(0)
bad = 0
for case in range(n_cases):
    vs = float(rng.choice([0.2, 0.35, 0.5, 1.0, 1.7]))
    cap = int(rng.choice([0, 3, 8, 20, 40]))
    mode = int(rng.choice([0, 0, 1]))
    n_scan = int(rng.choice([1500, 4000, 9000, 20000, 24000]))
    shift = float(rng.uniform(0.05, 1.2))
    match = str(rng.choice(["f", "q", "s", "f", "s", "o", "t", "w", "p", "f"]))  # f: the plan / scan matcher (default of large layers, round 5)
    if match in "otw" and not capi.dev_variants():  # (development matchers: tools/build_variants.sh)
        match = "f"
    seed = int(rng.integers(1, 10000))
    scene = synth.make_scene(seed, 60.0, 20)
    mp = synth.make_map(scene, int(rng.choice([40000, 150000])), seed)
    pose = [float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)), synth.SENSOR_H, float(rng.uniform(-0.2, 0.2)), 0.002, -0.002]
    scan = synth.make_scan(scene, pose, rings=48, azimuths=500, seed=seed + 1)[:n_scan]
    d = rng.normal(0, 1, 3)
    d *= shift / np.linalg.norm(d)
    guess = synth.pose_from_ypr(np.array(pose) + [d[0], d[1], 0.1 * d[2], float(rng.normal(0, 0.02)), 0.003, 0.002])
    iters = int(rng.choice([15, 40]))
    thr, kp = synth.threshold_schedule(2.0, iters)
    kw = dict(max_iterations=iters, threshold=thr, kernel_param=kp)
    if rng.integers(0, 3) == 0:  # converge all the way: the inner Gauss-Newton loop then ends early (step < min_delta)
        iters = 60
        thr, kp = synth.threshold_schedule(2.0, iters)
        kw = dict(max_iterations=iters, threshold=thr, kernel_param=kp, disable_stall_test=True)
    o = oracle_c.icp_align(oracle_c.Map(vs, cap, mode).insert(mp), scan, guess, oracle_c.ICPParams(**kw), want_pairs=True)
    os.environ["MH_MATCH"] = match
    g = capi.icp_align(capi.Map(ctx, vs, cap, mode).build(mp), capi.Scan(ctx, scan), guess, capi.ICPParams(**kw), want_pairs=True)
    ok = (g["n_iterations"] == o["n_iterations"] and g["termination_reason"] == o["termination_reason"] and
          [t["n_pairs"] for t in g["trace"]] == [t["n_pairs"] for t in o["trace"]] and
          all(np.array_equal(g["pairs"][k], o["pairs"][k]) for k in ("local_idx", "global_idx", "d2", "global_xyz")) and
          float(np.abs(g["T"] - o["T"]).max()) < 1e-9)
    bad += 0 if ok else 1
    print("case %2d vs=%.2f cap=%2d mode=%d n=%5d shift=%.2f match=%s iters=%d pairs=%d -> %s" % (
        case, vs, cap, mode, n_scan, shift, match, g["n_iterations"], g["n_final_pairs"], "ok" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
