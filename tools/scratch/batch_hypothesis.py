"""scratch: does one match launch over k scans' worth of points cost less per scan than k launches? (tail filling)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import capi, synth
w = synth.workload_c2()
ctx = capi.Context(0)
m = capi.Map(ctx, w.voxel_size, w.cap).build(w.map_xyz)
for k in (1, 2, 4, 8, 16):
    xyz = np.tile(w.scan_xyz, (k, 1))
    s = capi.Scan(ctx, xyz)
    p = capi.ICPParams(max_iterations=w.n_iters, disable_stall_test=True, threshold=w.threshold, kernel_param=w.kernel_param,
                       poll_every=w.n_iters, profile=1)
    for _ in range(2):
        capi.icp_align(m, s, w.T_guess, p)
    r = capi.icp_align(m, s, w.T_guess, p)
    t0 = time.perf_counter()
    for _ in range(3):
        r = capi.icp_align(m, s, w.T_guess, p)
    dt = (time.perf_counter() - t0) / 3
    print("k=%2d  n=%7d  match kernel %.1f us per launch = %.1f us per 120k points;  whole align %.2f ms = %.3f ms per scan-equivalent"
          % (k, len(xyz), 1e3 * r["match_kernel_ms"] / r["n_match_launches"], 1e3 * r["match_kernel_ms"] / r["n_match_launches"] / k,
             1e3 * dt, 1e3 * dt / k))
