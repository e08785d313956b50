#!/usr/bin/env python3
"""Accuracy of the device odometry against GROUND TRUTH on the synthetic city drive, with an ablation over every switch
whose upstream behaviour is unverified -- the one accuracy signal that does not pass through the oracle.

    python tools/accuracy_ablation.py --scans 1000 --out profiles/r04_accuracy.json

The drive (mola_lidar_odometry_amd/synth_city.py): 64 x 1875 rays, 80 m range, skewed sweeps with per-point time stamps in
the fourth float of a KITTI .bin row (molahip-lo-cli --time-field 12); a second folder holds the same drive with the sensor
standing still during every sweep (what KITTI's motion-compensated velodyne folders hold).  Every variant is ONE run of
molahip-lo-cli (C++, no Python in the loop) with environment variables the pipeline files / the host layer read.
Reported per variant: ATE RMSE [m] origin-aligned (the reference's tests, test/test_lidar_odometry_rawlog.cpp:95-104) and
SE(3)-aligned (evo_ape -a, eval/cli_mulran.sh:50), as % of the path, the end-point error, KITTI relative errors
(eval/cli_kitti.sh:41-50), rejected scans, scans/s.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import synth_city, trajectory  # noqa: E402

CLI = os.path.join(ROOT, "mola_lidar_odometry_amd", "molahip-lo-cli")
P_DEFAULT = os.path.join(ROOT, "pipelines", "lidar3d-default-hip.yaml")
P_NDT = os.path.join(ROOT, "pipelines", "lidar3d-ndt-hip.yaml")


def run_cli(seq_dir, pipeline, out, env=None, time_field=True, extra=()):
    cmd = [CLI, "--pipeline", pipeline, "--seq-dir", seq_dir, "--out", out] + (["--time-field", "12"] if time_field else []) + list(extra)
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=1200)
    if r.returncode != 0:
        raise RuntimeError("molahip-lo-cli failed: " + r.stderr[-500:])
    return next(json.loads(l) for l in r.stdout.splitlines() if l.startswith("{") and "sequence_dir" in l)


def score(rep, gt):
    _, est = trajectory.read_tum(rep["tum"])
    n = min(len(est), len(gt))
    path = synth_city.path_length(gt[:n, :3, :].reshape(n, 12)) if n > 1 else 0.0
    ate_o = trajectory.ate_rmse(est[:n], gt[:n], "origin")
    ate_s = trajectory.ate_rmse(est[:n], gt[:n], "se3")
    te, re_, nseg = trajectory.kitti_relative_errors(est[:n], gt[:n], lengths=(100, 200, 300, 400, 500))
    e0 = np.linalg.inv(est[0]) @ est[n - 1]
    g0 = np.linalg.inv(gt[0]) @ gt[n - 1]
    return {"ate_rmse_origin_m": ate_o, "ate_rmse_se3_m": ate_s, "path_m": path, "ate_origin_pct_of_path": 100.0 * ate_o / path if path else None,
            "end_error_m": float(np.linalg.norm(e0[:3, 3] - g0[:3, 3])), "kitti_t_err_pct": te, "kitti_r_err_deg_per_m": re_, "kitti_segments": nseg,
            "poses": len(est), "scans": rep["scans"], "good": rep["good"], "keyframes": rep["keyframes"],
            "icp_iterations_per_scan": rep["icp_iterations"] / max(1, rep["scans"]), "mean_icp_points": rep["mean_icp_points"],
            "final_map_points": rep["final_map_points"], "max_map_points": rep["max_map_points"],
            "steady_scans_per_s": rep["steady_scans_per_s"], "whole_run_scans_per_s": rep["scans_per_s"]}


VARIANTS = [
    # name, pipeline, env, folder ("skewed" | "still"), --time-field?
    ("default", P_DEFAULT, {}, "skewed", True),
    ("deskew_off", P_DEFAULT, {"MOLA_SKIP_DESKEW": "true"}, "skewed", True),
    ("no_time_stamps", P_DEFAULT, {}, "skewed", False),
    ("motion_compensated_input", P_DEFAULT, {}, "still", False),
    ("optimize_twist_off", P_DEFAULT, {"MOLA_OPTIMIZE_TWIST": "false"}, "skewed", True),
    ("motion_model_prior_on", P_DEFAULT, {"MOLA_HIP_MOTION_MODEL_PRIOR": "true"}, "skewed", True),
    ("gm_kiss", P_DEFAULT, {"MOLA_HIP_ROBUST_KERNEL": "GemanMcClure_KISS"}, "skewed", True),
    ("gm_barron", P_DEFAULT, {"MOLA_HIP_ROBUST_KERNEL": "GemanMcClure_Barron"}, "skewed", True),
    ("gm_c2", P_DEFAULT, {"MOLA_HIP_ROBUST_KERNEL": "GemanMcClure_C2"}, "skewed", True),
    ("cauchy", P_DEFAULT, {"MOLA_HIP_ROBUST_KERNEL": "Cauchy"}, "skewed", True),
    ("index_trunc", P_DEFAULT, {"MOLA_HIP_INDEX_MODE": "trunc"}, "skewed", True),
    ("far_voxel_l2", P_DEFAULT, {"MOLA_HIP_FAR_VOXEL_METRIC": "l2"}, "skewed", True),
    ("local_map_250m", P_DEFAULT, {"MOLA_LOCAL_MAP_MAX_SIZE": "250"}, "skewed", True),
    ("ndt_default", P_NDT, {}, "skewed", True),
    ("ndt_centroid_distance", P_NDT, {"MOLA_HIP_PT2PL_MODE": "centroid"}, "skewed", True),
    ("ndt_skip_plane_paired_points", P_NDT, {"MOLA_HIP_MATCHED_POINTS": "skip"}, "skewed", True),  # U12
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=1000)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_accuracy.json"))
    ap.add_argument("--only", default="", help="comma-separated variant names")
    ap.add_argument("--keep", default="", help="keep the generated sequence folders under this directory")
    args = ap.parse_args()
    only = set(v for v in args.only.split(",") if v)
    tmp_ctx = tempfile.TemporaryDirectory(prefix="molahip_acc_") if not args.keep else None
    base = args.keep or tmp_ctx.name
    t0 = time.time()
    need_still = any(v[3] == "still" and (not only or v[0] in only) for v in VARIANTS)
    d_skew, drive = synth_city.write_kitti_drive(os.path.join(base, "skewed"), args.scans, time_channel=True)
    d_still = synth_city.write_kitti_drive(os.path.join(base, "still"), args.scans, time_channel=False, skew=False)[0] if need_still else None
    gt = synth_city.ground_truth_44(drive)
    print("[ablation] drive of %d scans (%.0f m) generated in %.1f s" % (args.scans, synth_city.path_length(drive["poses"]), time.time() - t0),
          file=sys.stderr, flush=True)
    out = {"drive": {"scans": args.scans, "path_m": synth_city.path_length(drive["poses"]), "mean_points_per_scan": float(np.mean(drive["points_per_scan"])),
                     "generator": "mola_lidar_odometry_amd/synth_city.py (seed 2024): street grid, houses with gaps, parks, lots, cars, poles, trees; "
                                  "64 x 1875 rays, +2 .. -24.8 deg, 80 m, 2 cm range noise, 10 Hz, vehicle pulls away from rest, turns left and right"},
           "variants": {}}
    for name, pipe, env, folder, tf in VARIANTS:
        if only and name not in only:
            continue
        try:
            rep = run_cli(d_skew if folder == "skewed" else d_still, pipe, os.path.join(base, name + ".tum"), env, tf)
            out["variants"][name] = dict(score(rep, gt), env=env, pipeline=os.path.basename(pipe), input=folder, time_stamps=tf)
        except Exception as e:  # noqa: BLE001
            out["variants"][name] = {"error": repr(e)[:400]}
        v = out["variants"][name]
        print("[ablation] %-26s %s" % (name, "ATE %.3f m (%.3f %% of path), end %.2f m, good %d/%d, %.0f scans/s" % (
            v["ate_rmse_origin_m"], v["ate_origin_pct_of_path"], v["end_error_m"], v["good"], v["scans"], v["steady_scans_per_s"])
            if "error" not in v else v["error"]), file=sys.stderr, flush=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: (v.get("ate_origin_pct_of_path"), v.get("good")) for k, v in out["variants"].items()}))


if __name__ == "__main__":
    main()
