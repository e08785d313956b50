"""Several sequences on ONE GPU from ONE process (molahip-lo-cli with several --seq-dir) against the same sequence alone:
a host thread per sequence, alignments merged into lock-step batches (mp2p_icp_hip::AlignBatcher).
Writes the synthetic drive (HDL-64-like sweeps of ~120 k points) as a KITTI tree under a temporary directory once and points
N sequence folders at it.   python tools/multi_seq_bench.py [scans] [1,2,4,8,16] [pipeline.yaml]"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (generate_inputs: the drive's sweeps cast in worker processes)
from mola_lidar_odometry_amd import synth_city  # noqa: E402


def run(seq, n, pipeline, tmp):
    cmd = [bench.CLI, "--pipeline", pipeline, "--out", os.path.join(tmp, "t%d.tum" % n), "--profile", "--time-field", "12"]
    for _ in range(n):
        cmd += ["--seq-dir", seq]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        return {"error": r.stderr[-500:]}
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    per = [l for l in lines if "sequence_dir" in l]
    prof = [l["profile_ms_per_scan"] for l in lines if "profile_ms_per_scan" in l]
    summ = next((l for l in lines if "sequences" in l), None)
    keys = ("onLidar", "onLidar.1.filter_2nd", "onLidar.2.sensor_range", "onLidar.3.run_icp", "onLidar.4.update_local_map", "icp.host_polls")
    return {"steady_scans_per_s": (summ or per[0])["steady_scans_per_s"], "whole_run_scans_per_s": (summ or per[0])["scans_per_s"],
            "scans": sum(p["scans"] for p in per), "good": sum(p["good"] for p in per),
            "ms_per_scan_sequence_0": {k: round(prof[0][k], 4) for k in keys if prof and k in prof[0]},
            "tums": [p["tum"] for p in per]}


def main():
    n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    counts = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8, 16]
    pipeline = sys.argv[3] if len(sys.argv) > 3 else bench.PIPELINE
    tmp = tempfile.mkdtemp(prefix="molahip_multi_")
    seq, _ = synth_city.write_kitti_drive(tmp, n_scans, time_channel=True)  # the city drive of bench.py's extras
    out = {"pipeline": os.path.basename(pipeline), "scans_per_sequence": n_scans, "threads": {}}
    solo = None
    for mode in ("threads",):
        for n in counts:
            r = run(seq, n, pipeline, tmp)
            if "error" not in r:
                texts = [open(t).read() for t in r.pop("tums")]
                solo = solo or texts[0]
                r["trajectories_identical_to_solo_run"] = all(t == solo for t in texts)
            out[mode][str(n)] = r
            print(mode, n, json.dumps(r), flush=True)
    print(json.dumps({"multi_sequence_one_process": out}))


if __name__ == "__main__":
    main()
