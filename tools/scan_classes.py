"""Per-scan registration time of one sequence by ICP-layer size class (one-launch loop <= 2048 points | launch chain above) and by
iteration count: where the single sequence's time goes.   python tools/scan_classes.py [scans]"""
import json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mola_lidar_odometry_amd import synth_city  # noqa: E402
n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
tmp = tempfile.mkdtemp(prefix="molahip_sc_")
seq, _ = synth_city.write_kitti_drive(tmp, n_scans, time_channel=True)
out = os.path.join(tmp, "o.tum")
subprocess.run([bench.CLI, "--pipeline", bench.PIPELINE, "--seq-dir", seq, "--time-field", "12", "--out", out, "--scan-log", "auto"],
               capture_output=True, text=True, timeout=900)
d = np.genfromtxt(out[:-4] + "_scans.csv", delimiter=",", names=True)[5:]
res = {"scans": int(len(d)), "mean_ms": float(d["seconds"].mean() * 1e3), "scans_per_s": float(len(d) / d["seconds"].sum())}
for name, m in (("layer <= 2048 points (one-launch loop)", d["n_for_icp"] <= 2048), ("layer > 2048 points (launch chain)", d["n_for_icp"] > 2048)):
    s = d[m]
    if len(s):
        res[name] = {"share_of_scans": float(len(s) / len(d)), "mean_ms": float(s["seconds"].mean() * 1e3), "mean_iterations": float(s["icp_iterations"].mean()),
                     "mean_layer_points": float(s["n_for_icp"].mean()), "ms_per_iteration": float((s["seconds"] / np.maximum(1, s["icp_iterations"])).mean() * 1e3),
                     "share_of_time": float(s["seconds"].sum() / d["seconds"].sum())}
res["layer_points_percentiles_5_50_95_max"] = [float(np.percentile(d["n_for_icp"], q)) for q in (5, 50, 95, 100)]
res["iterations_percentiles_5_50_95_max"] = [float(np.percentile(d["icp_iterations"], q)) for q in (5, 50, 95, 100)]
res["align_calls_mean"] = float(d["align_calls"].mean())
print(json.dumps(res, indent=1))
