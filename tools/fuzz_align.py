"""development aid: random alignments with everything the solver side takes -- robust kernel forms, 1-4 inner Gauss-Newton
steps, min_delta / max_cost, priors, the device hook, stall thresholds, polling intervals, point maps and NDT maps
(Matcher_Point2Plane riding along) -- against the CPU oracle: iteration count, termination reason, pair counts per
iteration, final pairing indices bit for bit (d2 to the last fp32 bit), poses to 1e-7 (POSE_TOL of tests/test_gpu_parity.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import capi, synth  # noqa: E402
from oracle import oracle_c  # noqa: E402

oracle_c.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
ctx = capi.Context(0)
scene = synth.make_scene(31337, 70.0, 20)
bad = 0
for case in range(n_cases):
    ndt = bool(rng.integers(0, 3) == 0)
    n_scan = int(rng.choice([700, 1800, 3000, 7000, 15000, 40000]))
    seed = int(rng.integers(1, 10000))
    pose = [float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)), synth.SENSOR_H, float(rng.uniform(-0.2, 0.2)), 0.002, -0.002]
    cloud = synth.make_scan(scene, pose, rings=64, azimuths=1000, seed=seed)
    scan = cloud[rng.permutation(len(cloud))[:n_scan]]
    mp = synth.make_map(scene, int(rng.choice([60000, 150000])), seed)
    if ndt:
        margs = (float(rng.choice([1.0, 2.0])), int(rng.choice([0, 12])), 0, float(rng.choice([0.0, 0.1])), 0.05, 4)
    else:
        margs = (float(rng.choice([0.5, 1.0, 1.5])), int(rng.choice([0, 8, 20])), int(rng.integers(0, 2)))
    g_map, o_map = capi.Map(ctx, *margs).build(mp), oracle_c.Map(*margs).insert(mp)
    d = rng.normal(0, 1, 3)
    d *= float(rng.uniform(0.02, 0.6)) / np.linalg.norm(d)
    guess = synth.pose_from_ypr(np.array(pose) + [d[0], d[1], 0.1 * d[2], float(rng.normal(0, 0.01)), 0.002, 0.001])
    iters = int(rng.choice([5, 20, 50]))
    thr, kp = synth.threshold_schedule(float(rng.choice([0.5, 1.0, 2.0])), iters)
    inner = int(rng.choice([1, 2, 2, 4]))
    kernel = int(rng.integers(0, 6))
    gkw = dict(max_inner_iterations=inner, robust_kernel=kernel, min_delta=float(rng.choice([0.0, 1e-7, 1e-4])),
               max_cost=float(rng.choice([0.0, 0.0, 1e-3])))
    kw = dict(max_iterations=iters, threshold=thr, kernel_param=kp, disable_stall_test=bool(rng.integers(0, 3) == 0),
              min_abs_step_trans=float(rng.choice([1e-4, 5e-4])), min_abs_step_rot=float(rng.choice([1e-4, 5e-4])),
              threshold_angular_deg=float(rng.choice([0.0, 0.0, 0.5])))
    if ndt:
        kw["pt2pl_threshold"] = float(rng.choice([0.3, 0.6]))
    if rng.integers(0, 4) == 0:
        kw.update(hook_enabled=True, hook_min_trans=float(rng.choice([0.05, 0.2])), hook_min_rot=float(np.deg2rad(0.75)))
    prior = None
    if rng.integers(0, 4) == 0:
        prior = (guess, np.diag([20.0, 20.0, 20.0, 300.0, 300.0, 300.0]) * float(rng.choice([0.1, 1.0, 10.0])))
    poll = int(rng.choice([0, 0, 1, 3, 7, iters]))
    os.environ.pop("MH_MATCH", None)
    m = str(rng.choice(["", "", "q", "s", "p", "f"]))
    if m:
        os.environ["MH_MATCH"] = m
    try:
        a = capi.icp_align(g_map, capi.Scan(ctx, scan), guess, capi.ICPParams(gn=capi.GNParams(**gkw), poll_every=poll, **kw),
                           prior=prior, want_pairs=True)
        b = oracle_c.icp_align(o_map, scan, guess, oracle_c.ICPParams(gn=oracle_c.GNParams(**gkw), **kw), prior=prior, want_pairs=True)
        ok = (a["n_iterations"] == b["n_iterations"] and a["termination_reason"] == b["termination_reason"] and
              a["n_final_pairs"] == b["n_final_pairs"] and a.get("n_final_pairs_pt2pl", 0) == b.get("n_final_pairs_pt2pl", 0) and
              [t["n_pairs"] for t in a["trace"]] == [t["n_pairs"] for t in b["trace"]] and
              all(np.array_equal(a["pairs"][k], b["pairs"][k]) for k in ("local_idx", "global_idx")) and
              # d2 is fp32 arithmetic on a point transformed by the pose: poses agree to ~1e-13 (fp64 summation order), so a
              # coordinate on a rounding boundary may differ in its last bit (<= 8e-6 m at 100 m) and d2 by 2 |d| times that
              # -- anything more is a mismatch
              bool(np.all(np.abs(a["pairs"]["d2"] - b["pairs"]["d2"]) <= 2e-5 * np.sqrt(b["pairs"]["d2"]) + 1e-12)) and
              float(np.abs(a["T"] - b["T"]).max()) < 1e-7 and a["quality"] == b["quality"])
        note = "iters %d term %s pairs %d" % (a["n_iterations"], capi.TERM_NAMES[a["termination_reason"]], a["n_final_pairs"])
        if not ok:
            ta, tb = [t["n_pairs"] for t in a["trace"]], [t["n_pairs"] for t in b["trace"]]
            first = next((i for i, (x, y) in enumerate(zip(ta, tb)) if x != y), None)
            note += " | oracle iters %d term %s pairs %d; first differing iteration %s (%s vs %s); max |dT| %.3e; pairs equal %s; quality %r vs %r" % (
                b["n_iterations"], capi.TERM_NAMES[b["termination_reason"]], b["n_final_pairs"], first,
                ta[first] if first is not None else None, tb[first] if first is not None else None, float(np.abs(a["T"] - b["T"]).max()),
                [bool(np.array_equal(a["pairs"][k], b["pairs"][k])) for k in ("local_idx", "global_idx", "d2")], a["quality"], b["quality"])
            if len(a["pairs"]["d2"]) == len(b["pairs"]["d2"]) and len(a["pairs"]["d2"]):
                dd = np.abs(a["pairs"]["d2"] - b["pairs"]["d2"])
                k = int(dd.argmax())
                note += "; %d d2 values differ, worst %.3e at d2 = %.6e" % (int((dd > 0).sum()), float(dd[k]), float(b["pairs"]["d2"][k]))
            if first is not None and first > 0:
                note += "; |dT| before it %.3e" % float(np.abs(a["trace"][first - 1]["T"] - b["trace"][first - 1]["T"]).max())
    except capi.MolahipError as e:
        ok, note = False, "ERROR " + str(e)[-80:]
    bad += 0 if ok else 1
    print("case %3d ndt=%d n=%5d inner=%d kernel=%d min_delta=%g max_cost=%g stall_off=%d hook=%d prior=%d poll=%2d match=%-1s %s -> %s" % (
        case, ndt, n_scan, inner, kernel, gkw["min_delta"], gkw["max_cost"], kw["disable_stall_test"], "hook_enabled" in kw,
        prior is not None, poll, m, note, "ok" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
