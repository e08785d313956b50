"""development aid (CPU only): the two restatements against each other on random small alignments -- the C oracle every GPU
parity test leans on, and the independent float64 numpy restatement (expm / logm, dict-of-lists map, numpy.linalg.solve):
per-iteration pair counts, poses to 1e-9, termination, final pairings.  Voxel size, cap, index mode, robust kernel form,
1-4 inner steps, prior, stall test, guess error."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import synth  # noqa: E402
from oracle import icp_oracle_np as onp  # noqa: E402
from oracle import oracle_c  # noqa: E402

oracle_c.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 17)
scene = synth.make_scene(2718, 50.0, 14)
bad = 0
for case in range(n_cases):
    vs = float(rng.choice([0.5, 1.0, 1.7]))
    cap = int(rng.choice([0, 4, 20]))
    mode = int(rng.integers(0, 2))
    seed = int(rng.integers(1, 10000))
    mp = synth.make_map(scene, int(rng.choice([4000, 12000])), seed)
    pose = [float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)), synth.SENSOR_H, float(rng.uniform(-0.2, 0.2)), 0.002, -0.002]
    scan = synth.make_scan(scene, pose, rings=16, azimuths=200, seed=seed + 1)
    scan = scan[rng.permutation(len(scan))[:int(rng.choice([120, 300]))]]
    d = rng.normal(0, 1, 3)
    d *= float(rng.uniform(0.02, 0.5)) / np.linalg.norm(d)
    guess = synth.pose_from_ypr(np.array(pose) + [d[0], d[1], 0.1 * d[2], float(rng.normal(0, 0.01)), 0.002, 0.001])
    iters = int(rng.choice([4, 10, 25, 60]))  # (60 with the stall test off: the inner loop ends early once converged)
    thr, kp = synth.threshold_schedule(float(rng.choice([1.0, 2.0])), iters)
    inner = int(rng.choice([1, 2, 4]))
    kernel = int(rng.integers(0, 6))
    stall_off = bool(rng.integers(0, 2))
    prior = (guess, np.diag([20.0, 20.0, 20.0, 300.0, 300.0, 300.0]) * float(rng.choice([0.1, 1.0]))) if rng.integers(0, 3) == 0 else None
    a = oracle_c.icp_align(oracle_c.Map(vs, cap, mode).insert(mp), scan, guess,
                           oracle_c.ICPParams(max_iterations=iters, threshold=thr, kernel_param=kp, disable_stall_test=stall_off,
                                              gn=oracle_c.GNParams(max_inner_iterations=inner, robust_kernel=kernel)),
                           prior=prior, want_pairs=True)
    b = onp.icp_align(onp.VoxelMap(vs, cap, bool(mode)).insert(mp), scan, guess, thr, kp, iters, inner=inner, kernel=kernel,
                      disable_stall=stall_off, prior=prior)
    ok = a["n_iterations"] == b["n_iterations"] and oracle_c.TERM_NAMES[a["termination_reason"]] == b["termination_reason"]
    ok = ok and [t["n_pairs"] for t in a["trace"]][:len(b["trace"])] == [t["n_pairs"] for t in b["trace"]]
    worst = 0.0
    for ta, tb in zip(a["trace"], b["trace"]):
        worst = max(worst, float(np.abs(onp.T44(ta["T"]) - tb["T"]).max()))
    ok = ok and worst < 1e-9
    if b["pairs"] is not None and a["n_final_pairs"]:
        ok = ok and np.array_equal(a["pairs"]["global_idx"], b["pairs"]["global_idx"])
    bad += 0 if ok else 1
    print("case %3d vs=%.1f cap=%2d mode=%d n=%d inner=%d kernel=%d stall_off=%d prior=%d iters %d/%d term %s max|dT| %.1e -> %s" % (
        case, vs, cap, mode, len(scan), inner, kernel, stall_off, prior is not None, a["n_iterations"], b["n_iterations"],
        b["termination_reason"], worst, "ok" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
