"""scratch: wall_clock64 phase stamps of the one-workgroup alignment kernel k_icp_persist (debug library, tools/build_dbg.sh):
stamps of the LAST executed iteration -- 11 iteration start, 12 match done, 13 accumulated, 14 summed, 4..10 inside the solve
(see phase_probe.py), 15 solve returned."""
import ctypes as C, os, shutil, sys, time
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
m = capi.Map(ctx, w.voxel_size, w.cap).build(w.map_xyz[:200000])
for n in (400, 900, 2000):
    sel = rng.choice(len(w.scan_xyz), n, replace=False)
    s = capi.Scan(ctx, w.scan_xyz[sel])
    for iters in (1, 3, 6, 6):
        p = capi.ICPParams(max_iterations=iters, threshold=w.threshold[:iters], kernel_param=w.kernel_param[:iters], disable_stall_test=True,
                           gn=capi.GNParams(max_inner_iterations=1))
        t0 = time.perf_counter()
        capi.icp_align(m, s, w.T_guess, p, want_trace=False)
        dt = time.perf_counter() - t0
        buf = np.zeros(16, np.uint64)
        L.mh_debug_phases(buf.ctypes.data_as(C.c_void_p))
        t = buf.astype(np.int64)
        seq = [11, 12, 13, 14, 4, 5, 6, 7, 8, 9, 10, 15]
        print("n", n, "iters", iters, "wall %.1f us |" % (dt * 1e6), " ".join("%d->%d=%.2f" % (seq[i - 1], seq[i], (t[seq[i]] - t[seq[i - 1]]) / 100.0) for i in range(1, len(seq))),
              "| iteration %.2f us" % ((t[15] - t[11]) / 100.0))
