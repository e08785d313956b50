"""Where the time of N sequences in one process goes, seen from the device: runs molahip-lo-cli with N copies of the
synthetic drive under `rocprofv3 --kernel-trace` and reports, over the steady part of the run (the middle 60 % of the trace),
the share of wall time in which at least one kernel runs, the average number of kernels running at once, the busy share per
hardware queue, launches per scan and device time per scan by family (ICP loop / everything else).
    python tools/multi_seq_trace.py [scans] [n_sequences] [out.json]"""
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mola_lidar_odometry_amd import synth_city  # noqa: E402

ICP = ("k_match", "k_accum", "k_solve", "k_cov", "k_icp", "k_pairs", "k_compact")


def analyse(trace_dir, n_scans_total):
    rows = []
    for f in glob.glob(trace_dir + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    if not rows:
        return {"error": "no kernel trace"}
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo, hi = t0 + 0.2 * (t1 - t0), t0 + 0.8 * (t1 - t0)
    mid = [r for r in rows if lo <= r[0] < hi]
    ev = sorted([(r[0], 1) for r in mid] + [(r[1], -1) for r in mid])
    busy = 0
    area = 0
    depth = 0
    last = ev[0][0]
    for t, d in ev:
        if depth > 0:
            busy += t - last
            area += depth * (t - last)
        depth += d
        last = t
    wall = hi - lo
    fam = {"icp": [0, 0], "other": [0, 0]}
    names = {}
    for r in mid:
        short = r[2].split("(")[0].split("<")[0].replace("void ", "").replace("(anonymous namespace)::", "").replace("mh::", "")
        k = "icp" if any(short.startswith(p) for p in ICP) else "other"
        fam[k][0] += 1
        fam[k][1] += r[1] - r[0]
        n = names.setdefault(short[:40], [0, 0])
        n[0] += 1
        n[1] += r[1] - r[0]
    queues = {}
    for r in mid:
        queues[r[3]] = queues.get(r[3], 0) + (r[1] - r[0])
    # the lock-step alignment seen alone: durations of its kernels and the idle time between consecutive ones
    icp = [r for r in mid if any(r[2].replace("void ", "").startswith(p) for p in ("k_match", "k_accum", "k_solve"))]
    gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(icp, icp[1:])]
    small = [g for g in gaps if g < 15.0]
    mid_g = [g for g in gaps if 15.0 <= g < 150.0]
    icp_view = {"kernels": len(icp), "avg_kernel_us": sum(r[1] - r[0] for r in icp) / 1e3 / max(1, len(icp)),
                "gaps_below_15us": [len(small), sum(small) / max(1, len(small))],
                "gaps_15_to_150us": [len(mid_g), sum(mid_g) / max(1, len(mid_g))],
                "gaps_above_150us (between alignments)": [len(gaps) - len(small) - len(mid_g)]}
    # what runs beside the alignment's kernels: their duration by the family of the other kernels in flight when they start
    def family(name):
        n = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("mh::", "")
        if n.startswith("__amd_rocclr_copy"): return "copy"
        if n.startswith("__amd_rocclr_fill"): return "fill"
        if n.startswith("k_pp_deskew"): return "deskew"
        if n.startswith("k_pp_"): return "filters"
        if n.startswith("rocprim"): return "rocprim"
        if any(n.startswith(p) for p in ("k_match", "k_accum", "k_solve", "k_cov", "k_gather_states", "k_scatter_blocks")): return "icp"
        return "map_update"
    others = [(r[0], r[1], family(r[2])) for r in mid if family(r[2]) != "icp"]
    others.sort()
    beside = {}
    import bisect
    starts = [o[0] for o in others]
    for r in icp:
        k = bisect.bisect_right(starts, r[0])
        fams = sorted({o[2] for o in others[max(0, k - 40):k] if o[1] > r[0]})
        key = "+".join(fams) if fams else "alone"
        b = beside.setdefault(key, [0, 0])
        b[0] += 1
        b[1] += r[1] - r[0]
    icp_view["avg_kernel_us_by_what_runs_beside"] = {k: [v[0], round(v[1] / 1e3 / v[0], 2)] for k, v in sorted(beside.items(), key=lambda kv: -kv[1][0])}
    scans_mid = n_scans_total * 0.6
    return {"wall_ms_mid": wall / 1e6, "device_busy_share": busy / wall, "avg_kernels_running_when_busy": area / max(1, busy),
            "queues_busy_share": {q: round(v / wall, 3) for q, v in sorted(queues.items())},
            "alignment_kernels": icp_view, "launches_per_scan": {k: round(v[0] / scans_mid, 1) for k, v in fam.items()},
            "device_us_per_scan": {k: round(v[1] / 1e3 / scans_mid, 1) for k, v in fam.items()},
            "top_kernels_us_per_scan": {k: [round(v[0] / scans_mid, 2), round(v[1] / 1e3 / scans_mid, 1)]
                                        for k, v in sorted(names.items(), key=lambda kv: -kv[1][1])[:14]}}


def main():
    n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    n_seq = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    extra = sys.argv[4:]
    tmp = tempfile.mkdtemp(prefix="molahip_mtrace_")
    seq, _ = synth_city.write_kitti_drive(tmp, n_scans, time_channel=True)  # the city drive of bench.py's extras
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, "prof"), "--", bench.CLI, "--pipeline",
           bench.PIPELINE, "--out", os.path.join(tmp, "o.tum"), "--time-field", "12"] + extra
    for _ in range(n_seq):
        cmd += ["--seq-dir", seq]
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd="/tmp", env=env)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    summ = next((l for l in lines if "sequences" in l), None) or (lines[-1] if lines else {})
    rep = {"sequences": n_seq, "scans_per_sequence": n_scans, "under_rocprof_steady_scans_per_s": summ.get("steady_scans_per_s"),
           "trace": analyse(os.path.join(tmp, "prof"), n_seq * n_scans)}
    print(json.dumps(rep, indent=1))
    if out_path:
        json.dump(rep, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
