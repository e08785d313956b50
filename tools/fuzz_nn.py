"""development aid: the matcher-granular search (mh_nn_search_dense; mh_nn_search_k for pairingsPerPoint > 1) of every kernel family -- one lane per point (p, x),
quad, row, LDS tiles (t), wave-uniform candidates (w, with and without LDS staging) -- on random adversarial inputs against
the oracle's exhaustive 27-voxel scan: lattice-aligned maps and queries (exact ties across voxels), queries on voxel faces,
large coordinate offsets, awkward voxel sizes, caps, trunc indexing, ragged sizes.  Indices, d2 and records bit for bit."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import capi  # noqa: E402
from oracle import oracle_c  # noqa: E402

oracle_c.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 13)
ctx = capi.Context(0)
I12 = np.eye(4)[:3].reshape(12)
bad = 0
for case in range(n_cases):
    vs = float(rng.choice([0.25, 0.3, 0.5, 1.0, 1.7, 2.5]))
    cap = int(rng.choice([0, 1, 5, 20, 40]))
    mode = int(rng.choice([0, 0, 1]))
    offset = float(rng.choice([0.0, 0.0, 1000.0, -50000.0]))
    step = float(rng.choice([0.125, 0.25, 0.5]))
    g = np.arange(-6, 6, step, dtype=np.float32)
    lattice = np.stack(np.meshgrid(g, g, g[:max(4, len(g) // 2)], indexing="ij"), -1).reshape(-1, 3)
    lattice = lattice[rng.permutation(len(lattice))[:int(rng.choice([2000, 40000]))]]
    noise = rng.normal(0, float(rng.choice([1.0, 3.0, 10.0])), (int(rng.choice([100, 20000])), 3)).astype(np.float32)
    pts = (np.concatenate([lattice, noise]) + np.float32(offset)).astype(np.float32)
    n_q = int(rng.choice([1, 63, 700, 5000, 12000]))
    q_mid = (lattice[:n_q] + np.float32(step / 2)).astype(np.float32)
    q_bnd = np.round(rng.uniform(-6, 6, (n_q, 3)) / vs).astype(np.float32) * np.float32(vs)
    q_bnd[:, int(rng.integers(0, 3))] += rng.uniform(-0.5, 0.5, n_q).astype(np.float32)
    q_rnd = rng.uniform(-7, 7, (n_q, 3)).astype(np.float32)
    q_far = rng.uniform(-30, 30, (max(1, n_q // 8), 3)).astype(np.float32)
    q = (np.concatenate([q_mid, q_bnd, q_rnd, q_far]) + np.float32(offset)).astype(np.float32)
    q = q[rng.permutation(len(q))]
    # (a rotation about the origin would throw a cloud at 5e4 m hundreds of metres away: translations only there)
    xi = np.concatenate([rng.normal(0, 0.2, 3), rng.normal(0, 0.01, 3) if offset == 0.0 else np.zeros(3)])
    T = I12 if rng.integers(0, 2) else np.asarray(oracle_c.se3_exp(xi))
    om = oracle_c.Map(vs, cap, mode).insert(pts)
    o = oracle_c.match_points(om, q, T, 1e9)
    fails = []
    variants = (("p", {}), ("x", {}), ("q", {}), ("s", {}))
    if capi.dev_variants():  # (the development library: tools/build_variants.sh)
        variants += (("t", {}), ("w", {}), ("w", {"MH_WAVE_LDS": "1"}))
    for match, extra in variants:
        os.environ["MH_MATCH"] = match
        for k, v in extra.items():
            os.environ[k] = v
        gm = capi.Map(ctx, vs, cap, mode).build(pts)
        d = capi.nn_search_dense(gm, capi.Scan(ctx, q), T)
        for k in extra:
            del os.environ[k]
        found = d["global_idx"] != capi.NO_MATCH
        ok = (np.array_equal(np.nonzero(found)[0], o["local_idx"]) and np.array_equal(d["global_idx"][found], o["global_idx"]) and
              np.array_equal(d["d2"][found], o["d2"]) and np.array_equal(d["global_xyz"][found], o["global_xyz"]))
        if not ok:
            fails.append(match + ("+lds" if extra else ""))
    # Matcher_Points_DistanceThreshold with pairingsPerPoint = k (mh_nn_search_k) against the oracle's k-best restatement
    kk = int(rng.choice([2, 3, 5, 8]))
    thr_k = float(rng.choice([1e9, 0.4 * vs, 1.5 * vs]))
    ang_k = float(rng.choice([0.0, 0.0, 1.0]))
    ok_ = oracle_c.match_points_k(om, q, T, thr_k, kk, ang_k)
    os.environ["MH_MATCH"] = "q"
    dk = capi.nn_search_k(capi.Map(ctx, vs, cap, mode).build(pts), capi.Scan(ctx, q), T, thr_k, kk, ang_k)
    if not all(np.array_equal(dk[key], ok_[key]) for key in ("local_idx", "global_idx", "d2", "global_xyz")):
        fails.append("k%d" % kk)
    bad += 1 if fails else 0
    print("case %3d vs=%.2f cap=%2d mode=%d offset=%g step=%.3f map=%d queries=%d found=%d -> %s" % (
        case, vs, cap, mode, offset, step, len(pts), len(q), len(o["local_idx"]), "ok" if not fails else "MISMATCH " + ",".join(fails)), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
