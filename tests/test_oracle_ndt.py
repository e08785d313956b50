"""NDT voxel statistics + Matcher_Point2Plane restatement of the C oracle (SURVEY 8a row a13, App.B U10)
checked against numpy (eigh, brute force) and known answers.  lidar3d-ndt.yaml:195-200, 236-254."""
import numpy as np
import pytest

I12 = np.eye(4)[:3].reshape(12)


def wall_and_clutter(seed=0):
    rng = np.random.default_rng(seed)
    ground = np.stack([rng.uniform(-10, 10, 6000), rng.uniform(-10, 10, 6000), rng.normal(0.3, 0.01, 6000)], 1)
    wall = np.stack([rng.uniform(-10, 10, 4000), rng.normal(5.4, 0.01, 4000), rng.uniform(0.5, 4, 4000)], 1)
    blob = rng.normal([3.5, -3.5, 1.5], 0.25, (1500, 3))
    return np.concatenate([ground, wall, blob]).astype(np.float32)


def test_min_distance_insertion_filter(oracle):
    pts = np.array([[0.1, 0.1, 0.1], [0.15, 0.1, 0.1], [0.5, 0.1, 0.1], [0.5, 0.29, 0.1], [0.5, 0.31, 0.1],
                    [1.05, 0.1, 0.1]], np.float32)  # last one: other voxel, although < 0.2 from nothing there
    m = oracle.Map(1.0, 0, min_distance_between_points=0.2).insert(pts)
    d = m.dump()
    assert sorted(d["src_idx"].tolist()) == [0, 2, 4, 5]  # 1 too close to 0; 3 too close to 2 (0.19); 4 is 0.21 away
    assert oracle.Map(1.0, 0).insert(pts).num_points == 6


def test_ndt_statistics_match_numpy_eigh(oracle):
    pts = wall_and_clutter()
    m = oracle.Map(1.0, 0, min_distance_between_points=0.05, ndt_max_eigen_ratio=0.05, ndt_min_points=4).insert(pts)
    d, nd = m.dump(), m.dump_ndt()
    assert len(nd["is_plane"]) == m.num_voxels and nd["is_plane"].sum() > 50
    n_checked = 0
    for v in range(m.num_voxels):
        f, c = int(d["vox_first"][v]), int(d["vox_count"][v])
        P = d["xyz"][f:f + c].astype(np.float64)
        if c < 4:
            assert nd["is_plane"][v] == 0
            continue
        mu = P.mean(0)
        w, V = np.linalg.eigh(np.cov(P.T))  # ascending, 1/(n-1)
        np.testing.assert_allclose(nd["centroid"][v], mu.astype(np.float32), atol=1e-6)
        is_plane = w[2] > 0 and w[0] / w[2] < 0.05
        if abs(w[0] / max(w[2], 1e-300) - 0.05) > 1e-6:
            assert bool(nd["is_plane"][v]) == is_plane
        if is_plane and nd["is_plane"][v] and (w[1] - w[0]) > 1e-6 * w[2]:
            nrm = V[:, 0] * np.sign(V[np.argmax(np.abs(V[:, 0])), 0])
            np.testing.assert_allclose(nd["normal"][v], nrm, atol=2e-5)
            n_checked += 1
    assert n_checked > 50
    # ground voxels are planes with normal ~ +z, wall voxels with normal ~ +y
    keys = d["vox_keys"]
    g = (keys[:, 2] == 0) & (nd["is_plane"] == 1) & (np.abs(keys[:, 1] - 5) > 1)
    assert np.all(np.abs(nd["normal"][g][:, 2]) > 0.98)


def test_pt2pl_matcher_known_answer_and_bruteforce(oracle):
    pts = wall_and_clutter(1)
    m = oracle.Map(1.0, 0, min_distance_between_points=0.05, ndt_max_eigen_ratio=0.05).insert(pts)
    d, nd = m.dump(), m.dump_ndt()
    rng = np.random.default_rng(2)
    q = np.stack([rng.uniform(-9, 9, 500), rng.uniform(-9, 9, 500), rng.uniform(0.0, 0.8, 500)], 1).astype(np.float32)
    T = oracle.se3_exp([0.05, -0.03, 0.02, 0.004, -0.002, 0.003])
    r = oracle.match_pt2pl(m, q, T, 0.3)
    assert len(r["local_idx"]) > 300
    # brute force over the dumped voxel table
    key2v = {tuple(k): i for i, k in enumerate(d["vox_keys"].tolist())}
    Tm = np.asarray(T).reshape(3, 4)
    g = (q.astype(np.float64) @ Tm[:, :3].T + Tm[:, 3]).astype(np.float32)
    exp_idx, exp_c = [], []
    for i, p in enumerate(g):
        c = np.floor(p).astype(int)
        best, bv = np.inf, None
        for ix in (-1, 0, 1):
            for iy in (-1, 0, 1):
                for iz in (-1, 0, 1):
                    v = key2v.get((c[0] + ix, c[1] + iy, c[2] + iz))
                    if v is None or not nd["is_plane"][v]:
                        continue
                    dd = nd["centroid"][v] - p
                    d2 = np.float32(np.float32(dd[0] * dd[0] + dd[1] * dd[1]) + dd[2] * dd[2])
                    if d2 < best:
                        best, bv = d2, v
        if bv is not None:
            dd = p - nd["centroid"][bv]
            nn = nd["normal"][bv]
            e = np.float32(np.float32(nn[0] * dd[0] + nn[1] * dd[1]) + nn[2] * dd[2])
            if abs(e) < np.float32(0.3):
                exp_idx.append(i); exp_c.append(nd["centroid"][bv])
    np.testing.assert_array_equal(r["local_idx"], exp_idx)
    np.testing.assert_array_equal(r["centroid"], np.array(exp_c, np.float32))
    # SURVEY App. B U10, the other reading: distanceThreshold against the distance to the plane's CENTROID
    rc = oracle.match_pt2pl(m, q, T, 0.45, mode=oracle.PT2PL_CENTROID_DISTANCE)
    exp_c_idx = []
    for i, p in enumerate(g):
        c = np.floor(p).astype(int)
        best = np.inf
        for ix in (-1, 0, 1):
            for iy in (-1, 0, 1):
                for iz in (-1, 0, 1):
                    v = key2v.get((c[0] + ix, c[1] + iy, c[2] + iz))
                    if v is None or not nd["is_plane"][v]:
                        continue
                    dd = nd["centroid"][v] - p
                    d2 = np.float32(np.float32(dd[0] * dd[0] + dd[1] * dd[1]) + dd[2] * dd[2])
                    best = min(best, d2)
        if best < np.float32(0.45) * np.float32(0.45):
            exp_c_idx.append(i)
    np.testing.assert_array_equal(rc["local_idx"], exp_c_idx)
    assert 0 < len(exp_c_idx) < len(r["local_idx"])
    # points 0.3 m above the z=0.3 ground with a 0.1 m threshold pair with nothing from the ground
    hi = np.stack([rng.uniform(-9, 9, 50), rng.uniform(-9, 0, 50), np.full(50, 0.75)], 1).astype(np.float32)
    assert len(oracle.match_pt2pl(m, hi, I12, 0.1)["local_idx"]) == 0


def test_align_with_point2plane_recovers_offset(oracle):
    pts = wall_and_clutter(3)
    m = oracle.Map(1.0, 0, min_distance_between_points=0.05, ndt_max_eigen_ratio=0.05).insert(pts)
    rng = np.random.default_rng(4)
    scan = pts[rng.permutation(len(pts))[:3000]]
    guess = oracle.se3_exp([0.08, -0.06, 0.05, 0.004, -0.003, 0.006])
    base = dict(max_iterations=30, threshold=1.0, kernel_param=0.3, gn=oracle.GNParams(max_inner_iterations=1))
    a = oracle.icp_align(m, scan, guess, oracle.ICPParams(pt2pl_threshold=0.5, **base))
    assert a["n_final_pairs_pt2pl"] > 1000 and a["potential_pairings"] == 2 * len(scan)
    assert np.abs(a["T"] - I12).max() < 5e-3
    assert 0 < a["quality"] <= 1.0
    b = oracle.icp_align(m, scan, guess, oracle.ICPParams(**base))
    assert b["n_final_pairs_pt2pl"] == 0 and b["potential_pairings"] == len(scan)


def test_point_matcher_skips_plane_paired_points_when_asked():
    """U12 (added in round 4): with pt2pt_skip_plane_paired the point pairings of an iteration never share a
    local point with its plane pairings; without it (rounds 1-3) planar points carry both."""
    from mola_lidar_odometry_amd import synth
    from oracle import oracle_c as oc
    pts = synth.ndt_cloud(5)
    m = oc.Map(1.0, 0, 0, 0.1, 0.05, 4).insert(pts)
    rng = np.random.default_rng(6)
    scan = pts[rng.permutation(len(pts))[:3000]]
    guess = oc.se3_exp([0.08, -0.05, 0.04, 0.004, -0.003, 0.006])
    thr, kp = synth.threshold_schedule(0.5, 30)
    kw = dict(max_iterations=30, min_abs_step_trans=5e-4, min_abs_step_rot=5e-4, threshold=thr, kernel_param=kp, pt2pl_threshold=0.5,
              gn=oc.GNParams(max_inner_iterations=1))
    again = oc.icp_align(m, scan, guess, oc.ICPParams(**kw), want_pairs=True)
    skip = oc.icp_align(m, scan, guess, oc.ICPParams(pt2pt_skip_plane_paired=True, **kw), want_pairs=True)
    assert skip["n_final_pairs_pt2pl"] > 500 and again["n_final_pairs_pt2pl"] > 500
    n_skip = skip["n_final_pairs"] - skip["n_final_pairs_pt2pl"]
    n_again = again["n_final_pairs"] - again["n_final_pairs_pt2pl"]
    assert 0 < n_skip < n_again
    # the final pose's own matchers: no local point in both sets
    pl = oc.match_pt2pl(m, scan, skip["T"], 0.5)
    assert not set(skip["pairs"]["local_idx"].tolist()) & set(pl["local_idx"].tolist())
    assert set(again["pairs"]["local_idx"].tolist()) & set(oc.match_pt2pl(m, scan, again["T"], 0.5)["local_idx"].tolist())
    # both converge to the same place within the noise of the cloud
    assert np.abs(skip["T"] - again["T"]).max() < 2e-2
