#!/usr/bin/env python3
"""parity_pin.py -- the A/B that turns "parity vs restatement" into "parity vs reference" (last row).

Needs a box with the MOLA stack (mola-lidar-odometry-cli, mp2p_icp, MRPT) AND libmolahip + the adapter built against it
(mola_lidar_odometry_amd/host/adapters/CMakeLists.txt -> libmolahip_mp2p_icp.so); neither container of this project has
the stack, so this script has only been exercised against a stand-in executable (tests/test_parity_pin_tool.py).

What it does, for one dataset selection (everything after `--` goes to mola-lidar-odometry-cli unchanged, e.g.
`--input-kitti-seq 00`, with KITTI_BASE_DIR set as apps/mola-lidar-odometry-cli.cpp:255 expects):
  A. the reference:  mola-lidar-odometry-cli -c <ref pipeline> ... --output-tum-path A.tum          (its own CPU path)
  B. the drop-in:    mola-lidar-odometry-cli -l <plugin.so> -c <hip pipeline> ... --output-tum-path B_<variant>.tum
     once per candidate of the unverified upstream behaviours (U1 GemanMcClure weight form, the
     motion-model prior, ...), selected through the environment variables the -hip pipelines read;
  C. per variant: the largest per-scan translation / rotation difference between A and B; the variant within the
     north-star tolerance (1e-4 m, 1e-4 rad) pins the switches.  A.tum is copied next to the report as the reference-
     produced vector to commit under tests/golden/ (data, not code).
Both runs use (apps/mola-lidar-odometry-cli.cpp:93-95,553-562): -c/--config, -l/--load-plugins, --output-tum-path.
"""
import argparse
import itertools
import json
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import trajectory  # noqa: E402

SWITCHES = {  # environment variable -> candidates (first = this implementation's default)
    "MOLA_HIP_ROBUST_KERNEL": ["GemanMcClure", "GemanMcClure_KISS", "GemanMcClure_Barron"],
    "MOLA_HIP_MOTION_MODEL_PRIOR": ["false", "true"],
}
TOL_T, TOL_R = 1e-4, 1e-4


def run_cli(cli, args, env_extra, out_tum):
    env = dict(os.environ)
    env.update(env_extra)
    cmd = [cli] + args + ["--output-tum-path", out_tum]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(out_tum):
        raise RuntimeError("%s failed (%d):\n%s" % (" ".join(cmd), r.returncode, r.stderr[-2000:]))


def compare(a_tum, b_tum):
    sa, A = trajectory.read_tum(a_tum)
    sb, B = trajectory.read_tum(b_tum)
    ia, ib = trajectory.associate(sa, sb, max_dt=1e-3)
    if len(ia) == 0:
        return {"matched": 0, "max_dt_m": float("inf"), "max_dr_rad": float("inf")}
    A, B = A[ia], B[ib]
    dt = np.linalg.norm(A[:, :3, 3] - B[:, :3, 3], axis=1)
    R = np.einsum("nij,nkj->nik", A[:, :3, :3], B[:, :3, :3])
    dr = np.arccos(np.clip((np.trace(R, axis1=1, axis2=2) - 1) / 2, -1, 1))
    return {"matched": int(len(ia)), "scans_a": int(len(sa)), "scans_b": int(len(sb)), "max_dt_m": float(dt.max()),
            "max_dr_rad": float(dr.max()), "first_scan_over_tolerance": int(np.argmax((dt > TOL_T) | (dr > TOL_R)))
            if np.any((dt > TOL_T) | (dr > TOL_R)) else None}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--mola-cli", default="mola-lidar-odometry-cli")
    ap.add_argument("--plugin", required=True, help="libmolahip_mp2p_icp.so (the adapter built against the installed mp2p_icp)")
    ap.add_argument("--ref-pipeline", required=True, help="the reference's pipelines/lidar3d-default.yaml")
    ap.add_argument("--hip-pipeline", default=os.path.join(ROOT, "pipelines", "lidar3d-default-hip.yaml"))
    ap.add_argument("--out-dir", default="parity_pin_out")
    ap.add_argument("--only-defaults", action="store_true", help="no sweep: one B run with this implementation's defaults")
    ap.add_argument("dataset_args", nargs=argparse.REMAINDER, help="-- <arguments selecting the dataset, passed through>")
    args = ap.parse_args(argv)
    ds = [a for a in args.dataset_args if a != "--"]
    os.makedirs(args.out_dir, exist_ok=True)
    a_tum = os.path.join(args.out_dir, "A_reference.tum")
    run_cli(args.mola_cli, ["-c", args.ref_pipeline] + ds, {}, a_tum)
    names = list(SWITCHES)
    combos = [tuple(SWITCHES[n][0] for n in names)] if args.only_defaults else list(itertools.product(*[SWITCHES[n] for n in names]))
    rows = []
    for combo in combos:
        env = dict(zip(names, combo))
        tag = "_".join(combo)
        b_tum = os.path.join(args.out_dir, "B_%s.tum" % tag)
        run_cli(args.mola_cli, ["-l", args.plugin, "-c", args.hip_pipeline] + ds, env, b_tum)
        row = {"switches": env, "tum": b_tum}
        row.update(compare(a_tum, b_tum))
        row["within_tolerance"] = bool(row["max_dt_m"] <= TOL_T and row["max_dr_rad"] <= TOL_R)
        rows.append(row)
        print("%-60s max dt %.3e m  max dr %.3e rad  %s" % (tag, row["max_dt_m"], row["max_dr_rad"],
                                                            "PINNED" if row["within_tolerance"] else ""))
    rows.sort(key=lambda r: (r["max_dt_m"], r["max_dr_rad"]))
    report = {"tolerance": {"translation_m": TOL_T, "rotation_rad": TOL_R}, "reference_tum": a_tum, "dataset_args": ds,
              "best": rows[0], "pinned": bool(rows[0]["within_tolerance"]), "rows": rows,
              "next": "commit A_reference.tum + the dataset selection as tests/golden/ref_<dataset>.tum (data produced by the "
                      "reference) and make the winning switches the defaults of pipelines/*-hip.yaml"}
    json.dump(report, open(os.path.join(args.out_dir, "parity_pin_report.json"), "w"), indent=1)
    shutil.copy(a_tum, os.path.join(args.out_dir, "golden_ref.tum"))
    print(json.dumps({"pinned": report["pinned"], "best": report["best"]["switches"], "max_dt_m": report["best"]["max_dt_m"],
                      "max_dr_rad": report["best"]["max_dr_rad"]}))
    return 0 if report["pinned"] else 3


if __name__ == "__main__":
    sys.exit(main())
