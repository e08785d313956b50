"""development aid: sequences of DIFFERENT lengths (and both pipelines) through molahip-lo-cli in one process -- participants
leave the AlignBatcher at different times, batches shrink -- several times over; every trajectory must equal the solo run
of its sequence byte for byte."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import synth  # noqa: E402

CLI = os.path.join(ROOT, "mola_lidar_odometry_amd", "molahip-lo-cli")
base = "/tmp/molahip_stress"
lengths = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "12,19,26,33,40,47".split(","))]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
drive = synth.make_drive(max(lengths), rings=32, azimuths=600)
dirs = []
for n in lengths:
    d = os.path.join(base, "len%d" % n, "sequences", "00")
    os.makedirs(os.path.join(d, "velodyne"), exist_ok=True)
    for k in range(n):
        xyz = drive["scans"][k][0]
        np.concatenate([xyz, np.zeros((len(xyz), 1), np.float32)], 1).astype(np.float32).tofile(os.path.join(d, "velodyne", "%06d.bin" % k))
    np.savetxt(os.path.join(d, "times.txt"), (drive["stamps"] - drive["stamps"][0])[:n], fmt="%.6e")
    dirs.append(d)
bad = 0
for pipe in ("lidar3d-default-hip.yaml", "lidar3d-ndt-hip.yaml"):
    P = os.path.join(ROOT, "pipelines", pipe)
    solo = []
    for i, d in enumerate(dirs):
        out = os.path.join(base, "solo_%d.tum" % i)
        subprocess.run([CLI, "--pipeline", P, "--seq-dir", d, "--out", out], check=True, capture_output=True, timeout=600)
        solo.append(open(out).read())
    for r in range(rounds):
        order = np.random.default_rng(r).permutation(len(dirs))
        args = [CLI, "--pipeline", P, "--out", os.path.join(base, "multi.tum")]
        for i in order:
            args += ["--seq-dir", dirs[i]]
        res = subprocess.run(args, capture_output=True, text=True, timeout=900)
        same = res.returncode == 0 and all(open(os.path.join(base, "multi_%d.tum" % k)).read() == solo[i] for k, i in enumerate(order))
        bad += 0 if same else 1
        print("%s round %d order %s -> %s" % (pipe, r, [int(v) for v in order], "identical to the solo runs" if same else "MISMATCH " + res.stderr[-300:]), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
