"""scratch: per-wave begin/end times of one k_match4 launch (debug library built with -DMH_DEBUG_WAVETRACE)."""
import ctypes as C, os, shutil, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
shutil.copy(os.path.join(ROOT, "tools", "libmolahip_dbg.so"), os.path.join(ROOT, "mola_lidar_odometry_amd", "libmolahip.so"))
os.environ["MH_MATCH"] = os.environ.get("WT_VARIANT", "q")
os.environ["MH_NO_GRAPH"] = "1"
from mola_lidar_odometry_amd import capi, synth
w = synth.workload_c2()
L = capi.lib()
L.mh_debug_wavetrace.argtypes = [C.c_void_p, C.c_size_t]
L.mh_debug_wavetrace(None, 0)
ctx = capi.Context(0)
m = capi.Map(ctx, w.voxel_size, w.cap).build(w.map_xyz)
s = capi.Scan(ctx, w.scan_xyz)
nw = (4 * len(w.scan_xyz) + 63) // 64
out = {}
for reps in range(3):
    p = capi.ICPParams(max_iterations=1, threshold=w.threshold[:1], kernel_param=w.kernel_param[:1])
    capi.icp_align(m, s, w.T_guess, p)
    buf = np.zeros(2 * nw, np.uint64)
    L.mh_debug_wavetrace(buf.ctypes.data_as(C.c_void_p), nw)
    out["rep%d" % reps] = buf.reshape(-1, 2).copy()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "wavetrace_%s.npz" % os.environ["MH_MATCH"]), **out)
t = out["rep2"].astype(np.int64)
t0 = t[:, 0].min()
dur = (t[:, 1] - t[:, 0]) / 100.0  # wall_clock64 is 100 MHz -> us
print("variant", os.environ["MH_MATCH"], "stop", os.environ.get("MH_DBG_STOP", "0"), "waves", len(t), "kernel span us", (t[:, 1].max() - t0) / 100.0)
print("start offset us: pct", np.percentile((t[:, 0] - t0) / 100.0, [0, 50, 90, 99, 100]))
print("duration us: pct", np.percentile(dur, [0, 10, 50, 90, 99, 100]), "mean", dur.mean())
