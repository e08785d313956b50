"""development aid: rocprofv3 --hip-runtime-trace --stats over molahip-lo-cli with N sequences: host time per HIP API, per scan."""
import csv, glob, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from mola_lidar_odometry_amd import synth
n_scans = int(sys.argv[1]); n_seq = int(sys.argv[2]); extra = sys.argv[3:]
_, drive = bench.generate_inputs("small", [0], n_scans)
tmp = tempfile.mkdtemp(prefix="hipapi_")
seq = synth.write_kitti_sequence(tmp, drive)
cmd = ["rocprofv3", "--hip-runtime-trace", "--stats", "--output-format", "csv", "-d", os.path.join(tmp, "prof"), "--", bench.CLI,
       "--pipeline", bench.PIPELINE, "--out", os.path.join(tmp, "o.tum")] + extra
for _ in range(n_seq): cmd += ["--seq-dir", seq]
r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
print(json.dumps(next((l for l in lines if "sequences" in l), lines[-1] if lines else None)))
tot = n_scans * n_seq
for f in glob.glob(tmp + "/prof/**/*hip_api_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r_ in rows[:16]:
        print("%-34s calls/scan %6.2f  avg %8.2f us  total/scan %8.1f us" % (r_["Name"], int(r_["Calls"]) / tot, float(r_["AverageNs"]) / 1e3, float(r_["TotalDurationNs"]) / 1e3 / tot))
