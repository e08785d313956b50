"""The per-stage host profile (ms per scan, sequence 0) of N sequences in one molahip-lo-cli process.
    python tools/multi_seq_profile.py [scans] [n_sequences]   (environment is passed through: GPU_MAX_HW_QUEUES, MH_*)"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mola_lidar_odometry_amd import synth_city  # noqa: E402
n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_seq = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tmp = tempfile.mkdtemp(prefix="molahip_prof_")
seq, _ = synth_city.write_kitti_drive(tmp, n_scans, time_channel=True)
cmd = [bench.CLI, "--pipeline", bench.PIPELINE, "--out", os.path.join(tmp, "o.tum"), "--profile", "--time-field", "12"]
for _ in range(n_seq):
    cmd += ["--seq-dir", seq]
r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
prof = [l["profile_ms_per_scan"] for l in lines if "profile_ms_per_scan" in l]
summ = next((l for l in lines if "sequences" in l), None)
print(json.dumps({"summary": summ, "profile_ms_per_scan_seq0": {k: round(v, 4) for k, v in sorted(prof[0].items())} if prof else None}, indent=1))
print(r.stderr[-1500:])
