"""scratch: wall_clock64 phase stamps of k_accum_solve1 (debug library built with -DMH_DEBUG_WAVETRACE)."""
import ctypes as C, os, shutil, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
shutil.copy(os.path.join(ROOT, "tools", "libmolahip_dbg.so"), os.path.join(ROOT, "mola_lidar_odometry_amd", "libmolahip.so"))
os.environ["MH_NO_GRAPH"] = "1"
from mola_lidar_odometry_amd import capi, synth
rng = np.random.default_rng(0)
w = synth.workload_c2()
L = capi.lib()
ctx = capi.Context(0)
m = capi.Map(ctx, w.voxel_size, w.cap).build(w.map_xyz[:100000])
sel = rng.choice(len(w.scan_xyz), 900, replace=False)
s = capi.Scan(ctx, w.scan_xyz[sel])
names = ["start", "loads+acc", "wave_sum+lds", "partials+sync", "reduce_rows", "assemble", "ldlt", "exp+compose+store", "(inner bookkeeping)", "log", "tail"]
for iters in (1, 2, 3, 3):
    p = capi.ICPParams(max_iterations=iters, threshold=w.threshold[:iters], kernel_param=w.kernel_param[:iters], disable_stall_test=True)
    capi.icp_align(m, s, w.T_guess, p)
    buf = np.zeros(16, np.uint64)
    L.mh_debug_phases(buf.ctypes.data_as(C.c_void_p))
    t = buf.astype(np.int64)
    print("iters", iters, " ".join("%s=%.2f" % (names[i], (t[i] - t[i - 1]) / 100.0) for i in range(1, 11)), "total us %.2f" % ((t[10] - t[0]) / 100.0))
