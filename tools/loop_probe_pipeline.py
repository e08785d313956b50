"""The phases of the one-launch loop (k_icp16) inside the real pipeline: the first scans of the synthetic city drive through the
Python runner on the DEBUG library (phase stamps accumulated by workgroup 0 of the last alignment).

    bash tools/build_dbg.sh && mkdir -p /tmp/mhdbg && cp tools/libmolahip_dbg.so /tmp/mhdbg/libmolahip.so
    LD_LIBRARY_PATH=/tmp/mhdbg MOLAHIP_LIB_PATH=/tmp/mhdbg/libmolahip.so python tools/loop_probe_pipeline.py [scans]
"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import capi, synth_city
from mola_lidar_odometry_amd import _mp2p_icp_hip as H

n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 40
pipeline = os.path.join(ROOT, "pipelines", "lidar3d-default-hip.yaml")
L = capi.lib()
drive = synth_city.make_city_drive(n_scans)
lo = H.LidarOdometry(device=0, own_context=True)
lo.initialize(H.Config.FromYamlFile(pipeline))
names = ["(loop top)", "entries fetched", "ordered sums", "solve", "barrier after the body", "entries stored", "state + transform", "search", "accumulate"]
tot = np.zeros(13)
for k, ((xyz, t), st) in enumerate(zip(drive["scans"], drive["stamps"])):
    lo.onLidar(st, np.ascontiguousarray(xyz, dtype=np.float32), None if t is None else np.ascontiguousarray(t, dtype=np.float32))
    buf = np.zeros(32, np.uint64)
    L.mh_debug_phases(buf.ctypes.data_as(C.c_void_p))
    v = buf.astype(np.float64)[16:29]
    if k >= 5 and 0 < v[12] < 1000:
        tot += v
steps = max(1.0, tot[12])
if hasattr(L, "mh_debug_flat_counters") or True:
    try:
        fc = np.zeros(16, np.uint64)
        L.mh_debug_flat_counters(fc.ctypes.data_as(C.c_void_p), 0)
        fn = ["searches (waves)", "points", "unbounded at entry", "slow: no bound", "slow: > max candidates", "slow: chunk space", "slow: bound not attained", "candidates", "(unused)", "D rounds"]
        print("flat search counters: " + ", ".join("%s %d" % (fn[k], int(fc[k])) for k in range(10)))
    except Exception as e:
        print("no flat counters:", e)
print("scans %d..%d: %d steps | per step (us): %s | sum %.2f" % (5, n_scans - 1, int(steps), ", ".join("%s %.2f" % (names[k], tot[k] / 100.0 / steps) for k in (0, 1, 2, 3, 6, 7, 8, 4, 5)),
      tot[:9].sum() / 100.0 / steps))
