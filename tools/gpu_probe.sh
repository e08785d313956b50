#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python - <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")
from mola_lidar_odometry_amd import capi, synth
from test_gpu_parity import _ndt_cloud
from oracle import oracle_c
ctx = capi.Context(0)
pts = _ndt_cloud(11)
g = capi.Map(ctx, 1.0, 0, 0, 0.1, 0.05, 4).build(pts)
rng = np.random.default_rng(12)
for n in (1500, 5000, 20000):
    scan = pts[rng.permutation(len(pts))[:n]]
    s = capi.Scan(ctx, scan)
    guess = oracle_c.se3_exp([0.12, -0.09, 0.06, 0.006, -0.004, 0.01])
    iters = 20
    thr, kp = synth.threshold_schedule(0.5, iters)
    p = capi.ICPParams(max_iterations=iters, threshold=thr, kernel_param=kp, pt2pl_threshold=0.5, disable_stall_test=True, poll_every=iters)
    for _ in range(3): capi.icp_align(g, s, guess, p)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); capi.icp_align(g, s, guess, p); ts.append((time.perf_counter() - t0) * 1e6)
    print("NDT align n", len(scan), "per iter %.1f us" % (np.median(ts) / iters))
PY
