"""Generates tests/golden/ndt.json: the NDT / point-to-plane side of the path as the CPU
oracle computes it -- map statistics of a seeded cloud, one mixed point-to-plane + point-to-point alignment, and the
per-scan records of the lidar3d-ndt pipeline on the 12-scan synthetic drive.  Like the other fixtures it pins the
restatement and the product against drift, not the reference (parity unpinned).
Run from the repo root:  python tests/golden/make_golden_ndt.py"""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import synth  # noqa: E402
from oracle import odometry_oracle as oo  # noqa: E402
from oracle import oracle_c  # noqa: E402


def crc(a):
    return int(zlib.crc32(np.ascontiguousarray(a).tobytes()))


MAP = dict(voxel_size=1.0, cap=0, ndt=(0.1, 0.05, 4))  # min distance 0.1 m, eigen ratio 0.05, >= 4 points per plane
pts = synth.ndt_cloud(11)
m = oracle_c.Map(1.0, 0, 0, 0.1, 0.05, 4).insert(pts)
nd = m.dump_ndt()
rng = np.random.default_rng(12)
scan = pts[rng.permutation(len(pts))[:1500]]
guess = oracle_c.se3_exp([0.12, -0.09, 0.06, 0.006, -0.004, 0.01])
thr, kp = synth.threshold_schedule(0.5, 60)
al = oracle_c.icp_align(m, scan, guess, oracle_c.ICPParams(max_iterations=60, min_abs_step_trans=5e-4, min_abs_step_rot=5e-4,
                                                            threshold=thr, kernel_param=kp, pt2pl_threshold=0.5,
                                                            gn=oracle_c.GNParams(max_inner_iterations=2)))
d = synth.make_drive(12)
o = oo.OdometryOracle(os.path.join(ROOT, "pipelines", "lidar3d-ndt-hip.yaml"), n_threads=8)
recs = [o.on_lidar(st, s[0], s[1]) for s, st in zip(d["scans"], d["stamps"])]
out = {
    "cloud": {"seed": 11, "n": int(len(pts)), "crc": crc(pts)},
    "map": {"n_points": int(m.num_points), "n_voxels": int(m.num_voxels), "n_planes": int(nd["is_plane"].sum()),
            "plane_crc": crc(nd["is_plane"].astype(np.uint32)), "centroid_crc": crc(nd["centroid"])},
    "align": {"n_scan": 1500, "guess": [float(v) for v in guess], "n_iterations": int(al["n_iterations"]),
              "termination": oracle_c.TERM_NAMES[al["termination_reason"]], "n_final_pairs": int(al["n_final_pairs"]),
              "n_final_pairs_pt2pl": int(al["n_final_pairs_pt2pl"]), "T_final": [float(v) for v in al["T"]],
              "quality": float(al["quality"])},
    "drive": {"n_scans": 12, "pipeline": "pipelines/lidar3d-ndt-hip.yaml",
              "icp_iterations": [int(r["icp_iterations"]) for r in recs], "n_for_icp": [int(r["n_for_icp"]) for r in recs],
              "n_map_points": [int(r["n_map_points"]) for r in recs], "map_updated": [bool(r["map_updated"]) for r in recs],
              "poses": [[float(v) for v in np.asarray(r["pose"]).reshape(-1)] for r in recs]},
}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "ndt.json"), "w"), indent=1)
print("written", out["map"], out["align"]["n_iterations"], out["align"]["termination"], out["drive"]["icp_iterations"])
