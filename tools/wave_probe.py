"""scratch: per-tile times of one k_match_wave launch (debug library built with -DMH_DEBUG_WAVETRACE)."""
import ctypes as C, os, shutil, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
shutil.copy(os.path.join(ROOT, "tools", "libmolahip_dbg.so"), os.path.join(ROOT, "mola_lidar_odometry_amd", "libmolahip.so"))
os.environ["MH_MATCH"] = "w"
os.environ["MH_NO_GRAPH"] = "1"
from mola_lidar_odometry_amd import capi, synth
w = synth.workload_c2()
L = capi.lib()
L.mh_debug_wavetrace.argtypes = [C.c_void_p, C.c_size_t]
L.mh_debug_wavetrace(None, 0)
ctx = capi.Context(0)
m = capi.Map(ctx, w.voxel_size, w.cap).build(w.map_xyz)
s = capi.Scan(ctx, w.scan_xyz)
NT = 8192
for reps in range(3):
    p = capi.ICPParams(max_iterations=1, threshold=w.threshold[:1], kernel_param=w.kernel_param[:1])
    capi.icp_align(m, s, w.T_guess, p)
    buf = np.zeros(8 * NT, np.uint64)
    L.mh_debug_wavetrace(buf.ctypes.data_as(C.c_void_p), 4 * NT)
t = buf.reshape(-1, 8).astype(np.int64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
tot = (t[:, 1] - t[:, 0]) / 100.0
dense = t[:, 2] >= 40
print("tiles", len(t), "dense", int(dense.sum()), "kernel span us", (t[:, 1].max() - t0) / 100.0)
print("start offsets us pct(50,90,100)", np.percentile((t[:, 0] - t0) / 100.0, [50, 90, 100]).round(2))
for name, sel in (("dense", dense), ("sparse (wave 0's quad pass)", ~dense)):
    if sel.any():
        print(name, "total us pct(50,90,99,100)", np.percentile(tot[sel], [50, 90, 99, 100]).round(2), "mean", tot[sel].mean().round(2))
d = t[dense]
for i, nm in ((4, "bbox+probe+scan"), (5, "copy to LDS"), (6, "pass 1"), (1, "pass 2 + write")):
    prev = {4: 0, 5: 4, 6: 5, 1: 6}[i]
    ok = (d[:, i] > 0) & (d[:, prev] > 0)
    x = (d[ok, i] - d[ok, prev]) / 100.0
    print("  %-18s us pct(50,90,99,100) %s" % (nm, np.percentile(x, [50, 90, 99, 100]).round(2)))
print("  voxels pct", np.percentile(d[:, 3], [50, 90, 100]), "padded records pct", np.percentile(d[:, 7], [50, 90, 100]), "over LDS budget", int((d[:, 7] > 640).sum()), "over 128 voxels", int((d[:, 3] > 128).sum()))
worst = np.argsort(-tot)[:6]
for i in worst:
    print("tile", i, "points", t[i, 2], "total %.1f" % tot[i], "voxels", t[i, 3], "records", t[i, 7])
