"""Cost of one key-frame update (mh_map_insert) as a function of the stored map size: what the CALLER pays (the call returns
once the update is queued on the map's own stream) and when the update is complete (mh_map_get_info waits for its
counters)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mola_lidar_odometry_amd import capi, synth
w = synth.workload_c2()
ctx = capi.Context(0)
new = capi.Scan(ctx, w.scan_xyz[::12])  # 10 k points per key-frame
rows = {}
for n_map in (100_000, 250_000, 500_000, 1_000_000):
    m = capi.Map(ctx, 1.0, 20).build(w.map_xyz[:n_map])
    call, done = [], []
    for k in range(8):
        T = synth.pose_from_ypr([2.0 + 0.5 * k, -1.0, 1.7, 0.1, 0, 0])
        ctx.synchronize()
        t0 = time.perf_counter()
        m.insert(new, T, 150.0)
        t1 = time.perf_counter()
        n_now = m.info().n_points  # waits for the update's counters
        t2 = time.perf_counter()
        call.append(t1 - t0)
        done.append(t2 - t0)
    rows[n_map] = dict(call_ms=1e3 * float(np.median(call[2:])), complete_ms=1e3 * float(np.median(done[2:])), map_points=int(n_now))
    print("stored %8d points: key-frame update of %d points: call returns after %.3f ms, complete after %.3f ms (median of 6 "
          "after warm-up), map now %d points" % (n_map, len(w.scan_xyz[::12]), rows[n_map]["call_ms"], rows[n_map]["complete_ms"], n_now))
m = capi.Map(ctx, 1.0, 20).build(w.map_xyz); ctx.synchronize(); t0 = time.perf_counter()
m.build(w.map_xyz); ctx.synchronize()
full = 1e3 * (time.perf_counter() - t0)
print("full build of 1 M host points (incl. 12 MB upload): %.3f ms" % full)
print(json.dumps({"map_insert_10k_points": rows, "full_build_1M_ms": full}))
