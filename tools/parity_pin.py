#!/usr/bin/env python3
"""parity_pin.py -- the A/B that turns "parity vs restatement" into "parity vs reference" (last row).

Needs a box with the MOLA stack (mola-lidar-odometry-cli, mp2p_icp, MRPT, mola_metric_maps) AND libmolahip + the adapter
built against it (mola_lidar_odometry_amd/host/adapters/CMakeLists.txt -> libmolahip_mp2p_icp.so); neither container of
this project has the stack, so this script has only been exercised against a stand-in executable
(tests/test_parity_pin_tool.py).

For one dataset selection (everything after `--` goes to mola-lidar-odometry-cli unchanged, e.g. `--input-kitti-seq 00`,
with KITTI_BASE_DIR set as apps/mola-lidar-odometry-cli.cpp:255 expects):
  A.  the reference:  mola-lidar-odometry-cli -c <ref pipeline> ... --output-tum-path A.tum             (its own CPU path)
  A'. the reference THROUGH the plugin (MOLA_HIP_FORCE_CPU=1: every align() delegated to the upstream loop): must equal A,
      and leaves the reference's per-align record (nIterations, terminationReason, quality, pairing counts) in a CSV
  B.  the drop-in:    mola-lidar-odometry-cli -l <plugin.so> -c <-mola-hip pipeline> ... --output-tum-path B_<variant>.tum
      once per candidate of the unverified upstream behaviours, selected through the MOLA_HIP_* variables
      the adapter reads (molahip_host/plugin_switches.h); ONE switch is varied at a time around the defaults (--full-sweep:
      the cross product);
  C.  per variant: per-scan pose differences (TUM), per-scan sigma / sensor-range differences (the reference's own
      debug-traces CSV, LidarOdometry.cpp:2247-2282, which also carries time_onLidar = the authoritative CPU timing) and
      per-align {nIterations, quality, terminationReason} differences (MOLA_HIP_ALIGN_TRACE) -- written per scan to
      B_<variant>.per_scan.json, maxima in the report.  The variant within the north-star tolerance (1e-4 m, 1e-4 rad)
      pins the switches; A.tum is copied next to the report as the reference-produced vector to commit under
      tests/golden/ (data, not code).
The -mola-hip pipeline is the reference's own file with one class name changed; when --hip-pipeline is not given it is
derived on the spot from --ref-pipeline by pipelines/make_mola_hip.py.
Both runs use (apps/mola-lidar-odometry-cli.cpp:93-95,553-562): -c/--config, -l/--load-plugins, --output-tum-path.
"""
import argparse
import csv
import itertools
import json
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pipelines"))
from mola_lidar_odometry_amd import trajectory  # noqa: E402
import make_mola_hip  # noqa: E402  (pipelines/make_mola_hip.py -> lidar3d-default-mola-hip.yaml)

SWITCHES = {  # environment variable -> candidates (first = this implementation's default); SURVEY App. B id
    "MOLA_HIP_ROBUST_KERNEL": ["GemanMcClure", "GemanMcClure_KISS", "GemanMcClure_Barron", "GemanMcClure_C2"],  # U1
    "MOLA_HIP_INDEX_MODE": ["floor", "trunc"],                                                                 # U2/U3
    "MOLA_HIP_MIN_DELTA": ["1e-7", "1e-9", "1e-5"],                                                            # U8
    "MOLA_HIP_COV_STEP_XYZ": ["1e-7", "1e-6", "1e-8"],                                                         # U7 (with _ANG)
    "MOLA_HIP_PT2PL_MODE": ["plane", "centroid"],                                                              # U10 (NDT pipeline)
    "MOLA_HIP_FAR_VOXEL_METRIC": ["chebyshev", "l1", "l2"],                                                    # a8 (--device-map)
}
TOL_T, TOL_R = 1e-4, 1e-4


def run_cli(cli, args, env_extra, out_tum):
    env = dict(os.environ)
    env.update(env_extra)
    cmd = [cli] + args + ["--output-tum-path", out_tum]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(out_tum):
        raise RuntimeError("%s failed (%d):\n%s" % (" ".join(cmd), r.returncode, r.stderr[-2000:]))


def read_csv_columns(path):
    """{column: float array}; tolerant of the reference's trailing commas and quoted names (LidarOdometry.cpp:2274-2281)."""
    if not path or not os.path.exists(path):
        return {}
    with open(path, newline="") as f:
        rows = [[c for c in r] for r in csv.reader(f) if r]
    if len(rows) < 2:
        return {}
    names = [c.strip().strip('"') for c in rows[0]]
    cols = {}
    for j, n in enumerate(names):
        if not n:
            continue
        try:
            cols[n] = np.array([float(r[j]) if j < len(r) and r[j] != "" else np.nan for r in rows[1:]])
        except ValueError:
            cols[n] = [r[j] if j < len(r) else "" for r in rows[1:]]
    return cols


def compare(a_tum, b_tum, a_traces=None, b_traces=None, a_align=None, b_align=None):
    """-> (summary dict, per-scan list)."""
    sa, A = trajectory.read_tum(a_tum)
    sb, B = trajectory.read_tum(b_tum)
    ia, ib = trajectory.associate(sa, sb, max_dt=1e-3)
    if len(ia) == 0:
        return {"matched": 0, "max_dt_m": float("inf"), "max_dr_rad": float("inf")}, []
    A, B = A[ia], B[ib]
    dt = np.linalg.norm(A[:, :3, 3] - B[:, :3, 3], axis=1)
    R = np.einsum("nij,nkj->nik", A[:, :3, :3], B[:, :3, :3])
    dr = np.arccos(np.clip((np.trace(R, axis1=1, axis2=2) - 1) / 2, -1, 1))
    over = (dt > TOL_T) | (dr > TOL_R)
    out = {"matched": int(len(ia)), "scans_a": int(len(sa)), "scans_b": int(len(sb)), "max_dt_m": float(dt.max()),
           "max_dr_rad": float(dr.max()), "first_scan_over_tolerance": int(np.argmax(over)) if np.any(over) else None}
    per_scan = [{"stamp": float(sa[i]), "dt_m": float(dt[k]), "dr_rad": float(dr[k]), "pose_a": A[k][:3].reshape(-1).tolist(),
                 "pose_b": B[k][:3].reshape(-1).tolist()} for k, i in enumerate(ia)]
    # the reference's debug traces: one row per processed scan, all dynamic variables + time_onLidar
    ta, tb = read_csv_columns(a_traces), read_csv_columns(b_traces)
    for var in ("ADAPTIVE_THRESHOLD_SIGMA", "ESTIMATED_SENSOR_MAX_RANGE"):
        if var in ta and var in tb and len(ta[var]) == len(tb[var]) and len(ta[var]):
            d = np.abs(ta[var] - tb[var])
            out["max_abs_diff_" + var] = float(np.nanmax(d))
            for k in range(min(len(per_scan), len(d))):
                per_scan[k][var + "_a"], per_scan[k][var + "_b"] = float(ta[var][k]), float(tb[var][k])
    for tag, t in (("a", ta), ("b", tb)):
        if "time_onLidar" in t and len(t["time_onLidar"]):
            out["median_time_onLidar_s_" + tag] = float(np.nanmedian(t["time_onLidar"]))
    # per-align records written by the adapter (both loops): nIterations / quality / terminationReason
    ga, gb = read_csv_columns(a_align), read_csv_columns(b_align)
    if "nIterations" in ga and "nIterations" in gb:
        n = min(len(ga["nIterations"]), len(gb["nIterations"]))
        out["align_calls_a"], out["align_calls_b"] = int(len(ga["nIterations"])), int(len(gb["nIterations"]))
        if n:
            out["aligns_with_different_nIterations"] = int(np.sum(ga["nIterations"][:n] != gb["nIterations"][:n]))
            out["aligns_with_different_termination"] = int(np.sum(ga["terminationReason"][:n] != gb["terminationReason"][:n]))
            out["max_abs_diff_quality"] = float(np.max(np.abs(ga["quality"][:n] - gb["quality"][:n])))
            out["per_align"] = [{"call": k, "nIterations_a": int(ga["nIterations"][k]), "nIterations_b": int(gb["nIterations"][k]),
                                 "quality_a": float(ga["quality"][k]), "quality_b": float(gb["quality"][k]),
                                 "termination_a": int(ga["terminationReason"][k]), "termination_b": int(gb["terminationReason"][k])}
                                for k in range(n)]
    return out, per_scan


def variants(full):
    names = list(SWITCHES)
    defaults = tuple(SWITCHES[n][0] for n in names)
    if full:
        return names, list(itertools.product(*[SWITCHES[n] for n in names]))
    combos = [defaults]
    for i, n in enumerate(names):  # one switch at a time around the defaults
        for v in SWITCHES[n][1:]:
            combos.append(defaults[:i] + (v,) + defaults[i + 1:])
    return names, combos


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--mola-cli", default="mola-lidar-odometry-cli")
    ap.add_argument("--plugin", required=True, help="libmolahip_mp2p_icp.so (the adapter built against the installed mp2p_icp)")
    ap.add_argument("--ref-pipeline", required=True, help="the reference's pipelines/lidar3d-default.yaml (or lidar3d-ndt.yaml)")
    ap.add_argument("--hip-pipeline", default=None,
                    help="default: <out-dir>/<name>-mola-hip.yaml, e.g. lidar3d-default-mola-hip.yaml, derived from "
                         "--ref-pipeline by pipelines/make_mola_hip.py (one class name changed)")
    ap.add_argument("--device-map", action="store_true", help="derive the pipeline with mola::HashedVoxelPointCloudHIP as local map")
    ap.add_argument("--out-dir", default="parity_pin_out")
    ap.add_argument("--only-defaults", action="store_true", help="no sweep: one B run with this implementation's defaults")
    ap.add_argument("--full-sweep", action="store_true", help="cross product of all switches instead of one at a time")
    ap.add_argument("--skip-forced-cpu", action="store_true", help="skip run A' (the reference through the plugin)")
    ap.add_argument("dataset_args", nargs=argparse.REMAINDER, help="-- <arguments selecting the dataset, passed through>")
    args = ap.parse_args(argv)
    ds = [a for a in args.dataset_args if a != "--"]
    os.makedirs(args.out_dir, exist_ok=True)
    hip_pipeline = args.hip_pipeline
    if not hip_pipeline:
        text = open(args.ref_pipeline, encoding="utf-8").read()
        new, changes = make_mola_hip.transform(text, device_map=args.device_map)
        hip_pipeline = os.path.join(args.out_dir, os.path.basename(args.ref_pipeline).replace(".yaml", "-mola-hip.yaml"))
        with open(hip_pipeline, "w", encoding="utf-8") as f:
            f.write(new)
        print("derived %s from %s (changed lines: %s)" % (hip_pipeline, args.ref_pipeline, [c[0] for c in changes]))

    def traces_env(tag):
        return {"MOLA_SAVE_DEBUG_TRACES": "true", "MOLA_DEBUG_TRACES_FILE": os.path.join(args.out_dir, tag + ".traces.csv")}

    a_tum = os.path.join(args.out_dir, "A_reference.tum")
    run_cli(args.mola_cli, ["-c", args.ref_pipeline] + ds, traces_env("A_reference"), a_tum)
    a_traces = os.path.join(args.out_dir, "A_reference.traces.csv")
    a_align = None
    forced = None
    if not args.skip_forced_cpu:
        f_tum = os.path.join(args.out_dir, "A_forced_cpu.tum")
        a_align = os.path.join(args.out_dir, "A_forced_cpu.align.csv")
        env = dict(traces_env("A_forced_cpu"), MOLA_HIP_FORCE_CPU="1", MOLA_HIP_ALIGN_TRACE=a_align)
        run_cli(args.mola_cli, ["-l", args.plugin, "-c", hip_pipeline] + ds, env, f_tum)
        forced, _ = compare(a_tum, f_tum, a_traces, os.path.join(args.out_dir, "A_forced_cpu.traces.csv"))
        forced.pop("per_align", None)
        print("A' (reference through the plugin) vs A: max dt %.3e m, max dr %.3e rad" % (forced["max_dt_m"], forced["max_dr_rad"]))
    names, combos = variants(args.full_sweep)
    if args.only_defaults:
        combos = combos[:1]
    rows = []
    for combo in combos:
        env = dict(zip(names, combo))
        env["MOLA_HIP_COV_STEP_ANG"] = env["MOLA_HIP_COV_STEP_XYZ"]
        tag = "_".join(combo)
        b_tum = os.path.join(args.out_dir, "B_%s.tum" % tag)
        b_align = os.path.join(args.out_dir, "B_%s.align.csv" % tag)
        env.update(traces_env("B_" + tag))
        env["MOLA_HIP_ALIGN_TRACE"] = b_align
        run_cli(args.mola_cli, ["-l", args.plugin, "-c", hip_pipeline] + ds, env, b_tum)
        row = {"switches": dict(zip(names, combo)), "tum": b_tum}
        summary, per_scan = compare(a_tum, b_tum, a_traces, os.path.join(args.out_dir, "B_%s.traces.csv" % tag), a_align, b_align)
        per_align = summary.pop("per_align", None)
        json.dump({"switches": row["switches"], "per_scan": per_scan, "per_align": per_align},
                  open(os.path.join(args.out_dir, "B_%s.per_scan.json" % tag), "w"))
        row.update(summary)
        row["within_tolerance"] = bool(row["max_dt_m"] <= TOL_T and row["max_dr_rad"] <= TOL_R)
        rows.append(row)
        print("%-90s max dt %.3e m  max dr %.3e rad  %s" % (tag, row["max_dt_m"], row["max_dr_rad"],
                                                            "PINNED" if row["within_tolerance"] else ""))
    rows.sort(key=lambda r: (r["max_dt_m"], r["max_dr_rad"]))
    report = {"tolerance": {"translation_m": TOL_T, "rotation_rad": TOL_R}, "reference_tum": a_tum, "dataset_args": ds,
              "hip_pipeline": hip_pipeline, "reference_through_plugin": forced,
              "best": rows[0], "pinned": bool(rows[0]["within_tolerance"]), "rows": rows,
              "next": "commit A_reference.tum + the dataset selection as tests/golden/ref_<dataset>.tum (data produced by the "
                      "reference) and make the winning switches the defaults of molahip_host/plugin_switches.h and pipelines/*-hip.yaml"}
    json.dump(report, open(os.path.join(args.out_dir, "parity_pin_report.json"), "w"), indent=1)
    shutil.copy(a_tum, os.path.join(args.out_dir, "golden_ref.tum"))
    print(json.dumps({"pinned": report["pinned"], "best": report["best"]["switches"], "max_dt_m": report["best"]["max_dt_m"],
                      "max_dr_rad": report["best"]["max_dr_rad"]}))
    return 0 if report["pinned"] else 3


if __name__ == "__main__":
    sys.exit(main())
