"""Where one scan's wall clock goes on the device: molahip-lo-cli on N scans of the city drive under
`rocprofv3 --kernel-trace --memory-copy-trace`, then per alignment (one k_icp16 launch) the intervals around it.
    python tools/single_seq_timeline.py [scans]"""
import csv, glob, json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mola_lidar_odometry_amd import synth_city  # noqa: E402

n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tmp = tempfile.mkdtemp(prefix="molahip_tl_")
seq, _ = synth_city.write_kitti_drive(tmp, n_scans, time_channel=True)
cmd = ["rocprofv3", "--kernel-trace", "--memory-copy-trace", "--output-format", "csv", "-d", os.path.join(tmp, "prof"), "--", bench.CLI,
       "--pipeline", bench.PIPELINE, "--seq-dir", seq, "--time-field", "12", "--out", os.path.join(tmp, "o.tum")]
subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
ev = []
for f in glob.glob(tmp + "/prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")))
for f in glob.glob(tmp + "/prof/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", r.get("Name", "?"))))
ev.sort()
loops = [k for k, e in enumerate(ev) if e[2].startswith("k_icp16") or e[2].startswith("k_icpw")]
rows = []
for a, b in zip(loops[5:-1], loops[6:]):
    L = ev[a]
    seg = ev[a + 1:b]                 # everything between this loop and the next one
    prev = ev[a - 1]
    d = {"loop_us": (L[1] - L[0]) / 1e3, "prev_end_to_loop_start_us": (L[0] - prev[1]) / 1e3, "prev": prev[2],
         "loop_end_to_next_loop_start_us": (ev[b][0] - L[1]) / 1e3, "events_between": len(seg),
         "busy_between_us": sum(e[1] - e[0] for e in seg) / 1e3}
    cov = [e for e in seg if e[2].startswith("k_cov")]
    if cov:
        d["loop_end_to_cov_end_us"] = (cov[-1][1] - L[1]) / 1e3
    d2h = [e for e in seg if e[2].startswith("copy") and "DEVICE_TO_HOST" in e[2].upper()]
    if d2h:
        d["loop_end_to_first_d2h_end_us"] = (d2h[0][1] - L[1]) / 1e3
    rows.append(d)
keys = [k for k in rows[0] if k != "prev"]
out = {k: float(np.median([r[k] for r in rows if k in r])) for k in keys}
out["alignments"] = len(rows)
out["scan_period_us_median"] = float(np.median([ev[b][0] - ev[a][0] for a, b in zip(loops[5:-1], loops[6:])])) / 1e3
names = {}
for a, b in zip(loops[5:-1], loops[6:]):
    for e in ev[a + 1:b]:
        n = names.setdefault(e[2][:44], [0, 0.0]); n[0] += 1; n[1] += (e[1] - e[0]) / 1e3
out["between_two_loops_per_scan"] = {k: [round(v[0] / len(rows), 2), round(v[1] / len(rows), 1)] for k, v in sorted(names.items(), key=lambda kv: -kv[1][1])[:25]}
print(json.dumps(out, indent=1))
