"""tools/parity_pin.py is the A/B a maintainer runs on a box WITH the MOLA stack (reference CLI vs the same CLI with the
libmolahip adapter loaded).  That stack is absent here, so the tool is exercised against a stand-in executable that
understands the three options the tool relies on (-c, -l, --output-tum-path; apps/mola-lidar-odometry-cli.cpp:93-95) and
whose "plugin" run reproduces the "reference" run only for one setting of the switches: the sweep must find it."""
import json
import os
import stat
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STANDIN = r'''#!%s
import os, sys
a = sys.argv[1:]
out = a[a.index("--output-tum-path") + 1]
plugin = "-l" in a
kernel = os.environ.get("MOLA_HIP_ROBUST_KERNEL", "GemanMcClure")
prior = os.environ.get("MOLA_HIP_MOTION_MODEL_PRIOR", "false")
# the "reference" behaves like (GemanMcClure_KISS, prior on); every other plugin setting drifts by centimetres
off = 0.0 if (not plugin or (kernel == "GemanMcClure_KISS" and prior == "true")) else 0.03
with open(out, "w") as f:
    for k in range(20):
        f.write("%%.6f %%.9f 0 0 0 0 0 1\n" %% (0.1 * k, 0.7 * k + off * k))
''' % sys.executable


def test_sweep_finds_the_matching_variant(tmp_path):
    cli = tmp_path / "mola-lidar-odometry-cli"
    cli.write_text(STANDIN)
    cli.chmod(cli.stat().st_mode | stat.S_IEXEC)
    ref_yaml = tmp_path / "lidar3d-default.yaml"
    ref_yaml.write_text("params: {}\n")
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_pin.py"), "--mola-cli", str(cli), "--plugin", "libx.so",
                        "--ref-pipeline", str(ref_yaml), "--out-dir", str(out), "--", "--input-kitti-seq", "00"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.load(open(out / "parity_pin_report.json"))
    assert rep["pinned"] and rep["best"]["switches"] == {"MOLA_HIP_ROBUST_KERNEL": "GemanMcClure_KISS", "MOLA_HIP_MOTION_MODEL_PRIOR": "true"}
    assert sum(1 for row in rep["rows"] if row["within_tolerance"]) == 1 and len(rep["rows"]) == 6
    assert os.path.exists(out / "golden_ref.tum") and rep["dataset_args"] == ["--input-kitti-seq", "00"]
    # defaults only: not the matching variant -> exit code 3, report says so
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_pin.py"), "--mola-cli", str(cli), "--plugin", "libx.so",
                         "--ref-pipeline", str(ref_yaml), "--out-dir", str(tmp_path / "o2"), "--only-defaults", "--", "--input-kitti-seq", "00"],
                        capture_output=True, text=True, timeout=120)
    assert r2.returncode == 3 and not json.load(open(tmp_path / "o2" / "parity_pin_report.json"))["pinned"]
