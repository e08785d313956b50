"""scratch: cost of one key-frame update (mh_map_insert) as a function of the stored map size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mola_lidar_odometry_amd import capi, synth
w = synth.workload_c2()
ctx = capi.Context(0)
new = capi.Scan(ctx, w.scan_xyz[::12])  # 10 k points per key-frame
for n_map in (100_000, 250_000, 500_000, 1_000_000):
    m = capi.Map(ctx, 1.0, 20).build(w.map_xyz[:n_map])
    ts = []
    for k in range(6):
        T = synth.pose_from_ypr([2.0 + 0.5 * k, -1.0, 1.7, 0.1, 0, 0])
        t0 = time.perf_counter()
        m.insert(new, T, 150.0)
        ctx.synchronize()
        ts.append(time.perf_counter() - t0)
    print("stored %8d points: key-frame update of %d points takes %.3f ms (median of 5 after warm-up), map now %d points" % (
        n_map, len(w.scan_xyz[::12]), 1e3 * float(np.median(ts[1:])), m.info().n_points))
t0 = time.perf_counter(); m = capi.Map(ctx, 1.0, 20).build(w.map_xyz); ctx.synchronize(); t0 = time.perf_counter()
m.build(w.map_xyz); ctx.synchronize(); print("full build of 1 M host points (incl. 12 MB upload): %.3f ms" % (1e3 * (time.perf_counter() - t0)))
