"""Several sequences on ONE GPU from ONE process (molahip-lo-cli with several --seq-dir: a host thread per sequence, their
alignments merged into lock-step batches by mp2p_icp_hip::AlignBatcher) against the same sequences one after the other.
Writes the synthetic drive (HDL-64-like sweeps of ~120 k points) as a KITTI tree under /tmp once and points N sequence
folders at it."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import synth  # noqa: E402

CLI = os.path.join(ROOT, "mola_lidar_odometry_amd", "molahip-lo-cli")


def write_tree(root, drive):
    d = os.path.join(root, "sequences", "00")
    os.makedirs(os.path.join(d, "velodyne"), exist_ok=True)
    for k, (xyz, _) in enumerate(drive["scans"]):
        np.concatenate([xyz, np.zeros((len(xyz), 1), np.float32)], 1).astype(np.float32).tofile(os.path.join(d, "velodyne", "%06d.bin" % k))
    np.savetxt(os.path.join(d, "times.txt"), drive["stamps"] - drive["stamps"][0], fmt="%.6e")
    return d


def main():
    n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    counts = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8]
    pipeline = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "pipelines", "lidar3d-default-hip.yaml")
    t0 = time.time()
    drive = synth.make_drive(n_scans, rings=64, azimuths=1875)
    base = "/tmp/molahip_multi"
    seq = write_tree(base, drive)
    print("drive of %d scans, %d points each, written in %.1f s" % (n_scans, len(drive["scans"][0][0]), time.time() - t0), flush=True)
    out = {}
    for n in counts:
        args = [CLI, "--pipeline", pipeline, "--out", "/tmp/molahip_multi/out_%d.tum" % n]
        for _ in range(n):
            args += ["--seq-dir", seq]
        r = subprocess.run(args, capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            print("n =", n, "FAILED", r.stderr[-1500:])
            continue
        last = json.loads(r.stdout.strip().splitlines()[-1])
        per = [json.loads(l) for l in r.stdout.strip().splitlines() if l.startswith('{"sequence_dir"')]
        rate = last["scans_per_s"] if n > 1 else per[0]["scans_per_s"]
        steady = last["steady_scans_per_s"] if n > 1 else per[0]["steady_scans_per_s"]
        out[n] = {"scans_per_s": rate, "steady_scans_per_s": steady, "scans": sum(p["scans"] for p in per), "good": sum(p["good"] for p in per),
                  "icp_iterations": sum(p["icp_iterations"] for p in per),
                  "note": "wall clock incl. start-up for n > 1; registration time only (file reading excluded) for n = 1"}
        print("sequences in one process: %d -> %.0f scans/s whole run, %.0f steady state (registration time, first 5 scans left out)" % (n, rate, steady), flush=True)
    same = None  # (needs the solo run of THIS invocation)
    if 1 in out:
        same = all(open("/tmp/molahip_multi/out_%d_0.tum" % n).read() == open("/tmp/molahip_multi/out_1.tum").read() for n in counts if n > 1 and n in out)
    print(json.dumps({"multi_sequence_one_process": out, "trajectories_identical_to_solo_run": same}))


if __name__ == "__main__":
    main()
