"""development aid: N sequences in one molahip-lo-cli process, summary + sequence-0 stage table, with extra CLI flags / env."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from mola_lidar_odometry_amd import synth
n_scans = int(sys.argv[1]); counts = [int(v) for v in sys.argv[2].split(",")]
variants = sys.argv[3:] or [""]
_, drive = bench.generate_inputs("small", [0], n_scans)
tmp = tempfile.mkdtemp(prefix="ms_probe_")
seq = synth.write_kitti_sequence(tmp, drive)
for var in variants:
    flags = [f for f in var.split() if f.startswith("--")]
    env = dict(os.environ, **dict(kv.split("=") for kv in var.split() if "=" in kv and not kv.startswith("--")))
    for n in counts:
        cmd = [bench.CLI, "--pipeline", bench.PIPELINE, "--out", os.path.join(tmp, "o.tum"), "--profile"] + flags
        for _ in range(n): cmd += ["--seq-dir", seq]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        summ = next((l for l in lines if "sequences" in l), None)
        prof = [l["profile_ms_per_scan"] for l in lines if "profile_ms_per_scan" in l]
        print(repr(var), n, json.dumps(summ), json.dumps({k: round(v, 3) for k, v in prof[0].items() if k.startswith("onLidar")}) if prof else r.stderr[-300:], flush=True)
