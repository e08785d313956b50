"""development aid: random key-frame sequences (voxel size, cap, index mode, far-voxel radius, min distance, NDT) inserted into
a device map and into the CPU oracle's: voxel contents, source indices, bounding box and counts identical after every
insertion -- the merge path of mh_map_insert beyond the fixed cases of tests/test_gpu_preprocess.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import capi, synth  # noqa: E402
from oracle import oracle_c  # noqa: E402

oracle_c.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
ctx = capi.Context(0)
bad = 0
for case in range(n_cases):
    vs = float(rng.choice([0.3, 0.5, 1.0, 2.0]))
    cap = int(rng.choice([0, 2, 7, 20]))
    mode = int(rng.choice([0, 0, 1]))
    far = float(rng.choice([0.0, 12.0, 30.0, 60.0]))
    md = float(rng.choice([0.0, 0.0, 0.05, 0.2]))
    ndt = bool(rng.integers(0, 3) == 0)
    kw = dict(min_distance_between_points=md, ndt_max_eigen_ratio=0.03 if ndt else 0.0, ndt_min_points=4)
    seed = int(rng.integers(1, 10000))
    scene = synth.make_scene(seed, 70.0, 10)
    g = capi.Map(ctx, vs, cap, mode, md, kw["ndt_max_eigen_ratio"], 4)
    o = oracle_c.Map(vs, cap, mode, **kw)
    ok = True
    step = float(rng.uniform(2.0, 25.0))
    n_frames = int(rng.integers(3, 8))
    for k in range(n_frames):
        pose = [-30.0 + step * k, float(rng.uniform(-3, 3)), synth.SENSOR_H, 0.07 * k, 0.002 * k, -0.001 * k]
        xyz = synth.make_scan(scene, pose, rings=int(rng.choice([16, 32])), azimuths=int(rng.choice([200, 400])), seed=seed + k)
        if rng.integers(0, 6) == 0:
            xyz = xyz[:0]  # an empty key-frame
        T = synth.pose_from_ypr(pose)
        g.insert(capi.Scan(ctx, xyz), T, far)
        o.insert_posed(xyz, T, far)
        i = g.info()
        a, b = g.download(), o.dump()
        same = (i.n_points, i.n_voxels) == (o.num_points, o.num_voxels) and all(
            np.array_equal(a[key], b[key]) for key in ("vox_keys", "vox_first", "vox_count", "src_idx", "xyz"))
        if ndt and same and i.n_voxels:
            na, nb = g.download_ndt(), o.dump_ndt()
            same = np.array_equal(na["is_plane"], nb["is_plane"]) and np.allclose(na["centroid"], nb["centroid"], atol=1e-6)
        ok = ok and same
    bad += 0 if ok else 1
    print("case %2d vs=%.1f cap=%2d mode=%d far=%4.0f md=%.2f ndt=%d frames=%d points=%d -> %s" % (
        case, vs, cap, mode, far, md, ndt, n_frames, g.info().n_points, "ok" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
