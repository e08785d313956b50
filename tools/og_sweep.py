"""scratch: per-iteration cost of an alignment by layer size, with and without the fused row kernel (MH_NO_FUSE16)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import capi, synth
rng = np.random.default_rng(0)
w = synth.workload_c2()
ctx = capi.Context(0)
m = capi.Map(ctx, w.voxel_size, w.cap).build(w.map_xyz[:200000])
iters = 20
thr = np.full(iters, w.threshold[0]); kp = np.full(iters, w.kernel_param[0])
p = capi.ICPParams(max_iterations=iters, threshold=thr, kernel_param=kp, disable_stall_test=True, poll_every=iters)
for npts in (1000, 2000, 3000, 4000, 6000, 8000, 16000, 32000):
    s = capi.Scan(ctx, w.scan_xyz[rng.choice(len(w.scan_xyz), npts, replace=False)])
    row = []
    for nf in ("1", None):
        if nf: os.environ["MH_NO_FUSE16"] = nf
        else: os.environ.pop("MH_NO_FUSE16", None)
        for _ in range(3):
            capi.icp_align(m, s, w.T_guess, p)
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            capi.icp_align(m, s, w.T_guess, p)
            ts.append((time.perf_counter() - t0) * 1e6)
        row.append(np.median(ts))
    print("n", npts, "unfused %.0f us  fused %.0f us  per iter %.1f vs %.1f" % (row[0], row[1], row[0] / iters, row[1] / iters))
