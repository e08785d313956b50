python - <<PY
import os, sys, subprocess, tempfile, json
sys.path.insert(0, os.getcwd())
import bench
from mola_lidar_odometry_amd import synth_city
tmp = tempfile.mkdtemp(prefix="molahip_su_")
seq, _ = synth_city.write_kitti_drive(tmp, 400, time_channel=True)
for n in (1, 8):
    cmd = [bench.CLI, "--pipeline", bench.PIPELINE, "--out", os.path.join(tmp, "o%d.tum" % n), "--time-field", "12"]
    for _ in range(n): cmd += ["--seq-dir", seq]
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, MOLAHIP_STARTUP_LOG="1"))
    print("==== %d sequences" % n)
    print("\n".join(l for l in r.stderr.splitlines() if "startup" in l.lower() or "[su" in l.lower())[:3000])
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    s = next((l for l in lines if "sequences" in l), None) or lines[-1]
    print({k: s.get(k) for k in ("scans", "wall_seconds", "scans_per_s", "steady_scans_per_s")})
PY
