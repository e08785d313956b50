"""scratch: time of one small-layer alignment as a + b * iterations (fixed cost vs per-iteration cost)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import capi, synth
rng = np.random.default_rng(0)
w = synth.workload_c2()
ctx = capi.Context(0)
m = capi.Map(ctx, w.voxel_size, w.cap).build(w.map_xyz[:100000])
order = [int(a) for a in os.environ.get("FIT_ITERS", "1,2,5,10,20,40").split(",")]
for npts in (900, 4000):
    s = capi.Scan(ctx, w.scan_xyz[rng.choice(len(w.scan_xyz), npts, replace=False)])
    for iters in order:
        thr = np.full(iters, w.threshold[0]); kp = np.full(iters, w.kernel_param[0])
        for poll in (0, iters):
            p = capi.ICPParams(max_iterations=iters, threshold=thr, kernel_param=kp, disable_stall_test=True, poll_every=poll)
            for _ in range(3):
                capi.icp_align(m, s, w.T_guess, p)
            ts = []
            for _ in range(30):
                t0 = time.perf_counter()
                capi.icp_align(m, s, w.T_guess, p)
                ts.append((time.perf_counter() - t0) * 1e6)
            ts = np.array(ts)
            print("n", npts, "iters", iters, "poll", poll, "us per align: median %.1f mean %.1f max %.1f" % (np.median(ts), ts.mean(), ts.max()),
                  "per iter %.1f" % (np.median(ts) / iters))
