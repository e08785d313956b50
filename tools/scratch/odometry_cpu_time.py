"""scratch: wall time of the CPU oracle driver on the synthetic drive the GPU driver is profiled on."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import synth
from oracle import odometry_oracle as oo
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
d = synth.make_drive(n, rings=64, azimuths=1875)
o = oo.OdometryOracle(os.path.join(ROOT, "pipelines", "lidar3d-default-hip.yaml"), n_threads=threads)
t0 = time.perf_counter()
for (xyz, t), st in zip(d["scans"], d["stamps"]):
    o.on_lidar(st, xyz, t)
dt = time.perf_counter() - t0
print(json.dumps(dict(cpu_oracle_driver=True, scans=n, threads=threads, seconds=dt, scans_per_s=n / dt,
                      mean_points_raw=float(sum(len(s[0]) for s in d["scans"]) / n),
                      mean_points_for_icp=float(sum(r["n_for_icp"] for r in o.records) / n))))
