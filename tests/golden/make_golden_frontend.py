"""Generates tests/golden/frontend.json: golden vectors for SURVEY 8(f) rows f1-f3 on committed synthetic inputs --
filter index sets (as counts + checksums + head/tail samples), de-skewed coordinates, the local map after three
key-frame updates with far-voxel removal, and the per-scan output of the odometry driver on the 14-scan synthetic
drive.  Computed by the CPU oracle (oracle/icp_oracle.c, oracle/odometry_oracle.py); the reference itself cannot be
run here (parity unpinned), so these vectors pin the restatement against drift, not the reference.
Run from the repo root:  python tests/golden/make_golden_frontend.py"""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import synth  # noqa: E402
from oracle import odometry_oracle as oo  # noqa: E402
from oracle import oracle_c as oc  # noqa: E402


def crc(a):
    return int(zlib.crc32(np.ascontiguousarray(a).tobytes()))


drive = synth.make_drive(14)
xyz, t = drive["scans"][5]
PP = dict(decim_map_resolution=0.35, decim_icp_resolution=1.1, min_points_to_filter=300, range_min=2.0, range_max=70.0,
          bbox_mode=1, bbox_min=(-8.0, -8.0, -1.8), bbox_max=(8.0, 8.0, 4.0))
im, ii = oc.preprocess(xyz, **PP)
ta = oc.adjust_timestamps(t, oc.TS_MIDDLE_IS_ZERO, 0.0)
tw = [7.5, 0.02, 0.0, 0.0, 0.0, 0.11]
dsk = oc.deskew(xyz[ii], ta[ii], tw)

m = oc.Map(1.0, 20)
for k in (0, 4, 8):
    pts, _ = drive["scans"][k]
    m.insert_posed(pts[::3], drive["poses"][k], 60.0)
dump = m.dump()

o = oo.OdometryOracle(os.path.join(ROOT, "pipelines", "lidar3d-default-hip.yaml"), n_threads=8)
recs = [o.on_lidar(st, s[0], s[1]) for s, st in zip(drive["scans"], drive["stamps"])]

out = {
    "drive": {"n_scans": 14, "scan5_points": int(len(xyz)), "scan5_crc": crc(xyz), "scan5_t_crc": crc(t)},
    "preprocess": {"params": PP, "n_map": int(len(im)), "n_icp": int(len(ii)), "idx_map_crc": crc(im), "idx_icp_crc": crc(ii),
                   "idx_icp_head": [int(v) for v in ii[:8]], "idx_icp_tail": [int(v) for v in ii[-8:]]},
    "deskew": {"twist": tw, "xyz_crc_of_icp_layer": crc(dsk), "first3": [[float(v) for v in r] for r in dsk[:3]]},
    "map_insert": {"keyframes": [0, 4, 8], "stride": 3, "remove_voxels_farther_than": 60.0, "n_points": int(m.num_points),
                   "n_voxels": int(m.num_voxels), "xyz_crc": crc(dump["xyz"]), "src_crc": crc(dump["src_idx"]),
                   "keys_crc": crc(dump["vox_keys"]), "count_crc": crc(dump["vox_count"])},
    "odometry": {"pipeline": "pipelines/lidar3d-default-hip.yaml",
                 "poses": [[float(v) for v in r["pose"]] for r in recs],
                 "sigma": [float(r["sigma"]) for r in recs],
                 "icp_iterations": [int(r["icp_iterations"]) for r in recs],
                 "twist_corrections": [int(r["twist_corrections"]) for r in recs],
                 "map_updated": [bool(r["map_updated"]) for r in recs],
                 "n_for_icp": [int(r["n_for_icp"]) for r in recs],
                 "n_map_points": [int(r["n_map_points"]) for r in recs]},
}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "frontend.json"), "w"), indent=1)
print("written", out["preprocess"]["n_map"], out["preprocess"]["n_icp"], out["map_insert"]["n_points"], out["odometry"]["icp_iterations"])
