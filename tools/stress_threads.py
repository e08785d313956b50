"""development aid: the library is re-entrant across contexts -- T host threads, each with its own context, map and scans,
run alignments (and map insertions, filter chains) at the same time; every result must equal the one the same inputs give
on a single thread, bitwise."""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import capi, synth  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
scene = synth.make_scene(99, 70.0, 20)
mp = synth.make_map(scene, 120000, 99)
pose = [1.0, -0.5, synth.SENSOR_H, 0.04, 0.003, -0.002]
cloud = synth.make_scan(scene, pose, rings=64, azimuths=1000, seed=9)


def work(tid, out):
    rng = np.random.default_rng(1000 + tid)
    ctx = capi.Context(0)
    m = capi.Map(ctx, 1.0, 20).build(mp)
    res = []
    for r in range(reps):
        n = int(rng.choice([600, 2500, 9000, 30000]))
        s = capi.Scan(ctx, cloud[rng.permutation(len(cloud))[:n]])
        g = synth.pose_from_ypr(np.array(pose) + np.concatenate([rng.normal(0, 0.15, 3) * [1, 1, 0.1], rng.normal(0, 0.01, 3)]))
        iters = int(rng.choice([5, 20, 40]))
        thr, kp = synth.threshold_schedule(2.0, iters)
        a = capi.icp_align(m, s, g, capi.ICPParams(max_iterations=iters, threshold=thr, kernel_param=kp), want_trace=False, want_pairs=True)
        res.append((a["T"].copy(), a["cov"].copy(), a["n_iterations"], a["n_final_pairs"], a["pairs"]["global_idx"].copy()))
        if r % 7 == 3:  # a key-frame update and a filter chain in between
            m.insert(s, g, 60.0)
            om, oi = capi.Scan(ctx), capi.Scan(ctx)
            s.preprocess(capi.preprocess_params(0.5, 1.5, min_points_to_filter=100), om, oi)
            res.append((om.n, oi.n, int(m.info().n_points)))
    out[tid] = res


serial = {}
for t in range(T):
    work(t, serial)
par = {}
th = [threading.Thread(target=work, args=(t, par)) for t in range(T)]
for x in th:
    x.start()
for x in th:
    x.join(timeout=600)
bad = 0
for t in range(T):
    ok = t in par and len(par[t]) == len(serial[t]) and all(
        all(np.array_equal(u, v) for u, v in zip(a, b)) for a, b in zip(par[t], serial[t]))
    bad += 0 if ok else 1
    print("thread %d: %d results -> %s" % (t, len(serial[t]), "identical to the serial run" if ok else "MISMATCH"))
print("mismatches:", bad)
print("one-launch loops started / abandoned (whole process): %d / %d" % capi.loop_stats())
sys.exit(1 if bad else 0)
