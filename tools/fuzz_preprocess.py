"""development aid: random filter chains (decimation resolutions, minimum input size, index mode, range band and centre,
bounding-box modes, time-stamp adjustment, non-finite points) and de-skew twists against the CPU oracle: index sets,
coordinates and time stamps of both output layers bit for bit, de-skewed coordinates within one float ulp."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import capi, synth  # noqa: E402
from oracle import oracle_c  # noqa: E402

oracle_c.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 8)
ctx = capi.Context(0)
scene = synth.make_scene(555, 70.0, 20)
bad = 0
for case in range(n_cases):
    seed = int(rng.integers(1, 10000))
    pose = [float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)), synth.SENSOR_H, float(rng.uniform(-0.3, 0.3)), 0.0, 0.0]
    xyz = synth.make_scan(scene, pose, rings=int(rng.choice([16, 64])), azimuths=int(rng.choice([300, 1875])), seed=seed)
    xyz = (xyz - np.array(pose[:3], np.float32)).astype(np.float32)  # sensor frame
    n = min(len(xyz), int(rng.choice([len(xyz), len(xyz), 5000, 200, 0])))
    xyz = xyz[:n].copy()
    for _ in range(int(rng.integers(0, 4))):
        if n:
            xyz[int(rng.integers(0, n))] = [np.nan, 1, 1] if rng.integers(0, 2) else [1, -np.inf, 1]
    t = (1.7e4 + np.sort(rng.uniform(0.0, 0.1, n))).astype(np.float32)
    pp = dict(decim_map_resolution=float(rng.choice([0.0, 0.2, 0.35, 1.0])), decim_icp_resolution=float(rng.choice([0.0, 0.7, 1.1, 2.0])),
              min_points_to_filter=int(rng.choice([0, 300, 2000, 10 ** 7])), range_min=float(rng.choice([0.0, 2.0, 5.0])),
              range_max=float(rng.choice([0.0, 30.0, 70.0])), bbox_mode=int(rng.integers(0, 3)),
              bbox_min=(-float(rng.uniform(2, 9)), -float(rng.uniform(2, 9)), -1.8), bbox_max=(float(rng.uniform(2, 9)), float(rng.uniform(2, 9)), 4.0),
              range_center=(float(rng.choice([0.0, 0.5])), 0.0, 0.0),
              decim_map_method=int(rng.integers(0, 2)), decim_icp_method=int(rng.integers(0, 2)))  # FirstPoint | ClosestToAverage
    if rng.integers(0, 3) == 0 and n:  # points on a lattice: exact ties between a voxel's candidates
        xyz[::3] = np.round(xyz[::3] * 4) / 4
    mode = int(rng.integers(0, 2))
    ts = int(rng.integers(0, 3))
    off = float(rng.choice([0.0, 0.01]))
    with_t = bool(rng.integers(0, 4) != 0)
    ok, note = True, ""
    try:
        raw = capi.Scan(ctx, xyz)
        if with_t:
            raw.set_timestamps(t)
        om, oi = capi.Scan(ctx), capi.Scan(ctx)
        raw.preprocess(capi.preprocess_params(index_mode=mode, timestamp_method=ts, time_offset=off, **pp), om, oi)
        im, ii = oracle_c.preprocess(xyz, index_mode=mode, **pp)
        ta = oracle_c.adjust_timestamps(t, ts, off) if with_t else np.zeros(n, np.float32)
        for scan, idx in ((om, im), (oi, ii)):
            d = scan.download()
            ok = ok and scan.n == len(idx) and np.array_equal(d["src_idx"], idx) and np.array_equal(d["xyz"], xyz[idx], equal_nan=True)
            ok = ok and np.array_equal(d["t"], ta[idx] if with_t else np.zeros(len(idx), np.float32))
        note = "map %d icp %d" % (om.n, oi.n)
        if ok and om.n and with_t:
            tw = np.concatenate([rng.normal(0, 8, 3), rng.normal(0, 0.4, 3)])
            d0 = om.download()
            ref = oracle_c.deskew(d0["xyz"], d0["t"], tw)
            out = capi.Scan(ctx)
            om.deskew(tw, out)
            d = out.download()
            ulp = np.spacing(np.abs(ref).astype(np.float32))
            ok = ok and np.array_equal(d["src_idx"], d0["src_idx"]) and bool(np.all(np.abs(d["xyz"] - ref) <= ulp))
            note += " deskew max %.2e" % float(np.abs(d["xyz"] - ref).max())
    except capi.MolahipError as e:
        ok, note = False, "ERROR " + str(e)[-80:]
    bad += 0 if ok else 1
    print("case %3d n=%6d mode=%d ts=%d t=%d methods=%d%d %s -> %s" % (case, n, mode, ts, with_t, pp["decim_map_method"], pp["decim_icp_method"], note,
                                                                  "ok" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
