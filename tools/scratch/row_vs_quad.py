"""scratch: match-kernel time of the row (16 lanes/point) and quad (4 lanes/point) kernels vs layer size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from mola_lidar_odometry_amd import capi, synth
w = synth.workload_c2()
ctx = capi.Context(0)
m = capi.Map(ctx, 1.0, 20).build(w.map_xyz)
p = capi.ICPParams(max_iterations=w.n_iters, disable_stall_test=True, threshold=w.threshold, kernel_param=w.kernel_param, profile=1)
for n in (1000, 2000, 4000, 8000, 16000, 32000, 64000, 120000):
    sub = w.scan_xyz[np.linspace(0, len(w.scan_xyz) - 1, n).astype(int)]
    s = capi.Scan(ctx, sub)
    row = {}
    for v in ("s", "q"):
        os.environ["MH_MATCH"] = v
        capi.icp_align(m, s, w.T_guess, p)
        r = [capi.icp_align(m, s, w.T_guess, p) for _ in range(3)]
        row[v] = (1e3 * np.mean([x["match_kernel_ms"] / x["n_match_launches"] for x in r]), np.mean([x["total_ms"] for x in r]))
    print("n=%6d  match kernel: row16 %.1f us  quad %.1f us   whole align (20 it): row16 %.3f ms  quad %.3f ms" % (
        n, row["s"][0], row["q"][0], row["s"][1], row["q"][1]))
