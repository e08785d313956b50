"""scratch: per-tile phase times of one k_match_tile launch (debug library built with -DMH_DEBUG_WAVETRACE)."""
import ctypes as C, os, shutil, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
shutil.copy(os.path.join(ROOT, "tools", "libmolahip_dbg.so"), os.path.join(ROOT, "mola_lidar_odometry_amd", "libmolahip.so"))
os.environ["MH_MATCH"] = "t"
os.environ["MH_NO_GRAPH"] = "1"
from mola_lidar_odometry_amd import capi, synth
w = synth.workload_c2()
L = capi.lib()
L.mh_debug_wavetrace.argtypes = [C.c_void_p, C.c_size_t]
L.mh_debug_wavetrace(None, 0)
ctx = capi.Context(0)
m = capi.Map(ctx, w.voxel_size, w.cap).build(w.map_xyz)
s = capi.Scan(ctx, w.scan_xyz)
NT = 4096
for reps in range(3):
    p = capi.ICPParams(max_iterations=1, threshold=w.threshold[:1], kernel_param=w.kernel_param[:1])
    capi.icp_align(m, s, w.T_guess, p)
    buf = np.zeros(8 * NT, np.uint64)
    L.mh_debug_wavetrace(buf.ctypes.data_as(C.c_void_p), 4 * NT)
t = buf.reshape(-1, 8).astype(np.int64)
t = t[t[:, 0] > 0]
print("tiles", len(t))
t0 = t[:, 0].min()
print("kernel span us", (t[:, 5].max() - t0) / 100.0, "start offsets pct", np.percentile((t[:, 0] - t0) / 100.0, [50, 90, 100]))
names = ["bbox", "probe+scan", "copy", "search", "write"]
for i, nm in enumerate(names):
    d = (t[:, i + 1] - t[:, i]) / 100.0
    ok = (t[:, i + 1] > 0) & (t[:, i] > 0)
    print("%-12s us pct(50,90,99,100) %s mean %.2f" % (nm, np.percentile(d[ok], [50, 90, 99, 100]).round(2), d[ok].mean()))
tot = (t[:, 5] - t[:, 0]) / 100.0
print("tile total us pct", np.percentile(tot, [50, 90, 99, 100]).round(2), "mean", tot.mean().round(2))
nvox = t[:, 6]; rec = t[:, 7] & 0xFFFFFFFF; nocc = t[:, 7] >> 32
print("nvox pct", np.percentile(nvox, [50, 90, 100]), "records pct", np.percentile(rec, [50, 90, 100]), "occupied pct", np.percentile(nocc, [50, 90, 100]))
fb = (nvox > 256) | (rec > 1152)
print("fallback tiles", int(fb.sum()), "their total us", tot[fb].round(1)[:10])
worst = np.argsort(-tot)[:8]
for i in worst:
    print("tile", i, "total %.1f" % tot[i], "phases", ((t[i, 1:6] - t[i, 0:5]) / 100.0).round(1), "nvox", nvox[i], "rec", rec[i], "occ", nocc[i])
