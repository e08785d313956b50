"""development aid: per-stage host wall time of the odometry driver on the synthetic drive (LidarOdometry::profile())."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import run_odometry, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
drive = synth.make_drive(n, rings=64, azimuths=1875)
scans = [(float(t), xyz, ts) for (xyz, ts), t in zip(drive["scans"], drive["stamps"])]
recs, traj, dt = run_odometry.run_sequence(os.path.join(ROOT, "pipelines", "lidar3d-default-hip.yaml"), iter(scans), None)
last = recs[-1]
print("scans/s steady %.1f  iterations/scan %.2f" % (last["_steady_scans_per_s"], sum(r["icp_iterations"] for r in recs) / len(recs)))
for k, v in sorted(last["_host_ms_per_scan"].items()):
    print("  %-40s %.4f" % (k, v))
