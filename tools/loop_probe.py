"""Where a step of the one-launch loop (k_icp16) spends its time: accumulated wall_clock64 stamps of workgroup 0 (debug library:
tools/build_dbg.sh, -DMH_DEBUG_WAVETRACE).

    bash tools/build_dbg.sh && MOLAHIP_LIB_PATH=tools/libmolahip_dbg.so python tools/loop_probe.py
"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MOLAHIP_LIB_PATH", os.path.join(ROOT, "tools", "libmolahip_dbg.so"))
from mola_lidar_odometry_amd import capi, synth
rng = np.random.default_rng(0)
w = synth.workload_c2()
L = capi.lib()
ctx = capi.Context(0)
m = capi.Map(ctx, w.voxel_size, w.cap).build(w.map_xyz[rng.choice(len(w.map_xyz), 7000, replace=False)])
names = ["(loop top)", "entries fetched", "ordered sums", "solve", "barrier after the body", "entries stored", "state + transform", "search", "accumulate"]
for n in (700, 1400, 2000):
    s = capi.Scan(ctx, w.scan_xyz[rng.choice(len(w.scan_xyz), n, replace=False)])
    for rep in range(3):
        p = capi.ICPParams(max_iterations=40, threshold=w.threshold[:1].repeat(40), kernel_param=w.kernel_param[:1].repeat(40), disable_stall_test=True)
        t0 = time.perf_counter()
        r = capi.icp_align(m, s, w.T_guess, p, want_trace=False)
        dt = time.perf_counter() - t0
        buf = np.zeros(32, np.uint64)
        L.mh_debug_phases(buf.ctypes.data_as(C.c_void_p))
        t = buf.astype(np.float64)[16:]
        steps = max(1.0, t[12])
        print("n=%d iterations=%d steps=%d host %.1f us/step | per step (us): %s | sum %.2f" % (
            n, r["n_iterations"], int(steps), dt * 1e6 / steps, ", ".join("%s %.2f" % (names[k], t[k] / 100.0 / steps) for k in (0, 1, 2, 3, 6, 7, 8, 4, 5)),
            t[:9].sum() / 100.0 / steps), flush=True)
