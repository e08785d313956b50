"""wall_clock64 phase stamps of the small-layer kernels (debug library: tools/build_dbg.sh, -DMH_DEBUG_WAVETRACE; last launch wins).

    bash tools/build_dbg.sh && MOLAHIP_LIB_PATH=tools/libmolahip_dbg.so python tools/phase_probe.py

k_step16: the launch that finds the loop converged stamps start .. state written; the launch before it stamped
the body (search + sums); launches after the end only stamp `start`."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MOLAHIP_LIB_PATH", os.path.join(ROOT, "tools", "libmolahip_dbg.so"))
os.environ["MH_NO_GRAPH"] = "1"
from mola_lidar_odometry_amd import capi, synth
rng = np.random.default_rng(0)
w = synth.workload_c2()
L = capi.lib()
ctx = capi.Context(0)
m = capi.Map(ctx, w.voxel_size, w.cap).build(w.map_xyz[rng.choice(len(w.map_xyz), 7000, replace=False)])
names = {0: "start", 1: "state in LDS", 4: "reduce_rows", 5: "assemble", 6: "ldlt", 7: "exp+compose+store", 8: "(inner bookkeeping)", 9: "log", 10: "tail",
         11: "state written", 12: "search+accumulate", 13: "sums+partials"}
for n in (900, 1400, 2000):
    s = capi.Scan(ctx, w.scan_xyz[rng.choice(len(w.scan_xyz), n, replace=False)])
    for rep in range(3):
        p = capi.ICPParams(max_iterations=40, threshold=w.threshold[:1].repeat(40), kernel_param=w.kernel_param[:1].repeat(40))
        r = capi.icp_align(m, s, w.T_guess, p)
        buf = np.zeros(32, np.uint64)
        L.mh_debug_phases(buf.ctypes.data_as(C.c_void_p))
        t = buf.astype(np.int64)
        order = [k for k in (0, 1, 4, 5, 6, 7, 8, 9, 10, 11) if t[k]]
        line = " ".join("%s=%.2f" % (names[b], (t[b] - t[a]) / 100.0) for a, b in zip(order[:-1], order[1:]))
        body = " | body of the launch before: search+acc=%.2f sums+partials=%.2f" % ((t[12] - t[14]) / 100.0, (t[13] - t[12]) / 100.0) if t[13] else ""
        print("n=%d iterations=%d: %s total %.2f us%s" % (n, r["n_iterations"], line, (t[order[-1]] - t[order[0]]) / 100.0, body), flush=True)
