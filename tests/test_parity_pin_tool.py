"""tools/parity_pin.py is the A/B a maintainer runs on a box WITH the MOLA stack (reference CLI vs the same CLI with the
libmolahip adapter loaded).  That stack is absent here, so the tool is exercised against a stand-in executable that
understands the three options the tool relies on (-c, -l, --output-tum-path; apps/mola-lidar-odometry-cli.cpp:93-95), writes
the reference's debug-traces CSV (LidarOdometry.cpp:2247-2282) and the adapter's per-align CSV when asked through the
environment, and whose "plugin" run reproduces the "reference" run only for one setting of the switches: the sweep must
find it, per scan."""
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
cfg = open(a[a.index("-c") + 1]).read()
plugin = "-l" in a
assert plugin == ("mp2p_icp::ICP_HIP" in cfg), "the plugin run must get the -mola-hip pipeline, the reference run its own"
forced = os.environ.get("MOLA_HIP_FORCE_CPU") == "1"
kernel = os.environ.get("MOLA_HIP_ROBUST_KERNEL", "GemanMcClure")
index = os.environ.get("MOLA_HIP_INDEX_MODE", "floor")
# the "reference" behaves like (GemanMcClure_KISS, floor); every other plugin setting drifts by centimetres
same = (not plugin) or forced or (kernel == "GemanMcClure_KISS" and index == "floor")
off = 0.0 if same else 0.03
with open(out, "w") as f:
    for k in range(20):
        f.write("%%.6f %%.9f 0 0 0 0 0 1\n" %% (0.1 * k, 0.7 * k + off * k))
if os.environ.get("MOLA_SAVE_DEBUG_TRACES") == "true":
    with open(os.environ["MOLA_DEBUG_TRACES_FILE"], "w") as f:
        f.write('"ADAPTIVE_THRESHOLD_SIGMA","ESTIMATED_SENSOR_MAX_RANGE","time_onLidar","timestamp",\n')
        for k in range(20):
            f.write("%%f,%%f,%%f,%%f,\n" %% (2.0 - 0.05 * k + (0.0 if same else 0.01), 80.0, 0.02 if not plugin else 0.002, 0.1 * k))
if plugin and os.environ.get("MOLA_HIP_ALIGN_TRACE"):
    with open(os.environ["MOLA_HIP_ALIGN_TRACE"], "w") as f:
        f.write("call,loop,n_local,nIterations,terminationReason,quality,n_pt2pt,n_pt2pl,x,y,z,yaw,pitch,roll\n")
        for k in range(19):
            f.write("%%d,%%s,900,%%d,4,%%.6f,800,0,0,0,0,0,0,0\n" %% (k, "cpu" if forced else "hip", 25 if same else 26, 0.9 if same else 0.88))
''' % sys.executable


def test_sweep_finds_the_matching_variant(tmp_path):
    cli = tmp_path / "mola-lidar-odometry-cli"
    cli.write_text(STANDIN)
    cli.chmod(cli.stat().st_mode | stat.S_IEXEC)
    ref_yaml = tmp_path / "lidar3d-default.yaml"
    ref_yaml.write_text("params: {}\nicp_settings_with_vel:\n  class_name: mp2p_icp::ICP\n")
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_pin.py"), "--mola-cli", str(cli), "--plugin", "libx.so",
                        "--ref-pipeline", str(ref_yaml), "--out-dir", str(out), "--", "--input-kitti-seq", "00"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.load(open(out / "parity_pin_report.json"))
    # the -mola-hip pipeline was derived on the spot: the reference's text with the one class name changed
    assert rep["hip_pipeline"].endswith("lidar3d-default-mola-hip.yaml")
    assert open(rep["hip_pipeline"]).read() == ref_yaml.read_text().replace("mp2p_icp::ICP", "mp2p_icp::ICP_HIP")
    assert rep["pinned"] and rep["best"]["switches"]["MOLA_HIP_ROBUST_KERNEL"] == "GemanMcClure_KISS"
    assert rep["best"]["switches"]["MOLA_HIP_INDEX_MODE"] == "floor"
    # one switch at a time around the defaults: 1 + (3 + 1 + 2 + 2 + 1 + 2) runs, exactly one within tolerance
    assert len(rep["rows"]) == 12 and sum(1 for row in rep["rows"] if row["within_tolerance"]) == 1
    assert os.path.exists(out / "golden_ref.tum") and rep["dataset_args"] == ["--input-kitti-seq", "00"]
    # the reference through the plugin (MOLA_HIP_FORCE_CPU=1) reproduces the reference
    assert rep["reference_through_plugin"]["max_dt_m"] == 0.0
    # per-scan / per-align detail: traces, timing, iteration counts
    best = rep["best"]
    assert best["max_abs_diff_ADAPTIVE_THRESHOLD_SIGMA"] == 0.0 and best["aligns_with_different_nIterations"] == 0
    assert best["median_time_onLidar_s_a"] == 0.02 and best["median_time_onLidar_s_b"] == 0.002
    worst = rep["rows"][-1]
    assert worst["aligns_with_different_nIterations"] == 19 and abs(worst["max_abs_diff_quality"] - 0.02) < 1e-9
    assert abs(worst["max_abs_diff_ADAPTIVE_THRESHOLD_SIGMA"] - 0.01) < 1e-6
    detail = json.load(open(out / ("B_%s.per_scan.json" % "_".join(worst["switches"].values()))))
    assert len(detail["per_scan"]) == 20 and len(detail["per_align"]) == 19
    assert detail["per_scan"][5]["dt_m"] > 0.1 and "ADAPTIVE_THRESHOLD_SIGMA_b" in detail["per_scan"][5]
    assert detail["per_align"][0] == {"call": 0, "nIterations_a": 25, "nIterations_b": 26, "quality_a": 0.9, "quality_b": 0.88,
                                      "termination_a": 4, "termination_b": 4}
    # defaults only: not the matching variant -> exit code 3, report says so
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_pin.py"), "--mola-cli", str(cli), "--plugin", "libx.so",
                         "--ref-pipeline", str(ref_yaml), "--out-dir", str(tmp_path / "o2"), "--only-defaults", "--skip-forced-cpu",
                         "--", "--input-kitti-seq", "00"], capture_output=True, text=True, timeout=120)
    assert r2.returncode == 3 and not json.load(open(tmp_path / "o2" / "parity_pin_report.json"))["pinned"]
