"""Stress for the several-sequences-in-one-process runner: sequences of DIFFERENT lengths and drives (so that they leave the
batcher at different times, re-align at different scans and sit out filter rounds), default and NDT
pipeline, repeated; every trajectory must be byte-identical to the sequence's solo run and no run may hang.
    python tools/stress_cli_sequences.py [repeats]"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mola_lidar_odometry_amd import synth  # noqa: E402

NDT = os.path.join(ROOT, "pipelines", "lidar3d-ndt-hip.yaml")


def main():
    repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    tmp = tempfile.mkdtemp(prefix="molahip_stress_")
    dirs = []
    for k, (n, seed, speed) in enumerate(((9, 11, 6.0), (17, 22, 9.0), (26, 33, 12.0), (13, 44, 4.0), (31, 55, 14.0))):
        d = synth.make_drive(n, seed=seed, speed=speed)
        dirs.append(synth.write_kitti_sequence(os.path.join(tmp, "s%d" % k), d))
    bad = 0
    runs = 0
    for pipe in (bench.PIPELINE, NDT):
        solo = []
        for k, d in enumerate(dirs):
            out = os.path.join(tmp, "solo_%d.tum" % k)
            r = subprocess.run([bench.CLI, "--pipeline", pipe, "--seq-dir", d, "--out", out], capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr
            solo.append(open(out).read())
        for rep in range(repeats):
            for mode in ([],):
                order = dirs[rep % len(dirs):] + dirs[:rep % len(dirs)]
                ref = solo[rep % len(dirs):] + solo[:rep % len(dirs)]
                out = os.path.join(tmp, "m.tum")
                cmd = [bench.CLI, "--pipeline", pipe, "--out", out] + mode
                for d in order:
                    cmd += ["--seq-dir", d]
                runs += 1
                try:
                    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
                except subprocess.TimeoutExpired:
                    print("HANG", os.path.basename(pipe), mode, rep, flush=True)
                    bad += 1
                    continue
                same = r.returncode == 0 and all(open(os.path.join(tmp, "m_%d.tum" % k)).read() == ref[k] for k in range(len(order)))
                summ = [j for j in (json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")) if "batches" in j]
                print(os.path.basename(pipe), mode or ["threads"], rep, "ok" if same else "DIFFERENT " + r.stderr[-200:],
                      {k: summ[0][k] for k in ("batches", "filter_batches", "filter_jobs", "filter_timeouts")} if summ else "", flush=True)
                bad += 0 if same else 1
    print("runs: %d  mismatches: %d" % (runs, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
