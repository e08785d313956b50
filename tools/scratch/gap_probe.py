"""development aid: per scan, the device-idle gaps around the alignment in a single-sequence molahip-lo-cli run (kernel trace)."""
import csv, glob, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from mola_lidar_odometry_amd import synth
n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 60
_, drive = bench.generate_inputs("small", [0], n_scans)
tmp = tempfile.mkdtemp(prefix="gap_")
seq = synth.write_kitti_sequence(tmp, drive)
cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, "prof"), "--", bench.CLI, "--pipeline", bench.PIPELINE,
       "--out", os.path.join(tmp, "o.tum"), "--seq-dir", seq]
subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
rows = []
for f in glob.glob(tmp + "/prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("mh::", "").split("(")[0].split("<")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "")))
rows.sort()
# the main stream's queue = the one k_match16 runs on
q = next(r[3] for r in rows if r[2].startswith("k_match16"))
main = [r for r in rows if r[3] == q]
def med(v):
    v = sorted(v); return v[len(v) // 2] / 1e3 if v else None
g = {"icp_end_to_insert_start": [], "insert_span": [], "insert_end_to_deskew_start": [], "deskew_end_to_match_start": [], "icp_span": []}
i = 0
names = [r[2] for r in main]
for k, r in enumerate(main):
    if r[2] == "k_pp_deskew_pair":
        # next k_match16 after it
        nxt = next((m for m in main[k + 1:] if m[2].startswith("k_match16")), None)
        if nxt: g["deskew_end_to_match_start"].append(nxt[0] - r[1])
        prev = main[k - 1] if k else None
        if prev: g["insert_end_to_deskew_start"].append(r[0] - prev[1])
    if r[2] in ("k_init_build",) and k:
        g["icp_end_to_insert_start"].append(r[0] - main[k - 1][1])
        end = next((m for m in main[k:] if m[2] == "k_table_insert"), None)
        if end: g["insert_span"].append(end[1] - r[0])
first = None
for k, r in enumerate(main):
    if r[2].startswith("k_match16") and (k == 0 or not (main[k - 1][2].startswith("k_match16") or main[k - 1][2].startswith("k_accum_solve1"))):
        first = r[0]
    if r[2].startswith("k_cov_finalize") and first:
        g["icp_span"].append(r[1] - first); first = None
print(json.dumps({k: {"median_us": med(v), "n": len(v)} for k, v in g.items()}, indent=1))
print("main-queue kernels by name:", sorted({n for n in names}))
