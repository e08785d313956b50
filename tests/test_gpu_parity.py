"""-m gpu: the HIP path, called through the C ABI, against the CPU oracle on the same seeded inputs.

Bars (north_star): bit-exact for the integer/index work (voxel contents, NN indices, fp32 d^2,
pair counts, termination); fp64 solver quantities to ~1e-9 relative; final poses far inside the
1e-4 m / 1e-4 rad tolerance (asserted at 1e-7)."""
import numpy as np
import pytest

from mola_lidar_odometry_amd import capi, synth

pytestmark = pytest.mark.gpu

I12 = np.eye(4)[:3].reshape(12)
POSE_TOL = 1e-7  # north_star allows 1e-4 m / 1e-4 rad; we hold the same iterates to 1e-7


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def small(ctx, oracle, small_workload):
    w = small_workload
    gm = capi.Map(ctx, w.voxel_size, w.cap).build(w.map_xyz)
    om = oracle.Map(w.voxel_size, w.cap).insert(w.map_xyz)
    gs = capi.Scan(ctx, w.scan_xyz)
    return w, gm, om, gs


def assert_maps_equal(g, o):
    np.testing.assert_array_equal(g["vox_keys"], o["vox_keys"])
    np.testing.assert_array_equal(g["vox_first"], o["vox_first"])
    np.testing.assert_array_equal(g["vox_count"], o["vox_count"])
    np.testing.assert_array_equal(g["src_idx"], o["src_idx"])
    np.testing.assert_array_equal(g["xyz"], o["xyz"])


# ---------------------------------------------------------------------------- map build
@pytest.mark.parametrize("vs,cap,mode", [(1.0, 20, 0), (0.5, 3, 0), (1.7, 0, 0), (1.0, 20, 1), (0.25, 1, 0)])
def test_map_build_bit_exact(ctx, oracle, vs, cap, mode):
    rng = np.random.default_rng(int(vs * 100) + cap)
    pts = np.concatenate([rng.normal(0, 6, (30000, 3)), rng.uniform(-0.5, 0.5, (5000, 3))]).astype(np.float32)
    pts[17] = [np.nan, 0, 0]
    pts[99] = [0, np.inf, 0]
    g = capi.Map(ctx, vs, cap, mode).build(pts)
    o = oracle.Map(vs, cap, mode).insert(pts)
    i = g.info()
    assert (i.n_points, i.n_voxels, i.n_offered) == (o.num_points, o.num_voxels, len(pts))
    assert_maps_equal(g.download(), o.dump())
    mn, mx = o.bbox()
    np.testing.assert_array_equal(np.array(i.bbox_min), mn)
    np.testing.assert_array_equal(np.array(i.bbox_max), mx)
    assert i.table_size >= 2 * i.n_voxels and (i.table_size & (i.table_size - 1)) == 0


def test_map_rebuild_and_empty(ctx, oracle):
    g = capi.Map(ctx, 1.0, 20)
    assert g.info().n_points == 0 and g.info().n_voxels == 0
    g.build(np.random.default_rng(0).normal(0, 3, (1000, 3)))
    n1 = g.info().n_points
    g.build(np.zeros((0, 3), np.float32))
    assert g.info().n_points == 0
    pts = np.random.default_rng(1).normal(0, 3, (2000, 3)).astype(np.float32)
    g.build(pts)
    assert_maps_equal(g.download(), oracle.Map(1.0, 20).insert(pts).dump())
    assert n1 > 0


@pytest.mark.parametrize("voxel,knn,min_pts,trunc", [(0.5, 10, 6, False), (1.0, 5, 3, False), (0.4, 16, 8, False), (0.5, 10, 6, True)])
def test_point2plane_matcher_on_a_plain_point_map_knn_pca(ctx, oracle, voxel, knn, min_pts, trunc):
    """Matcher_Point2Plane where the global layer has no plane statistics (pipelines/rgbd.yaml:143-151: a HashedVoxelPointCloud
    layer; SURVEY 8a row a13 "otherwise KNN + PCA"): mh_nn_search_pt2pl_knn against the oracle's restatement -- the same index
    set, centroids and normals to rounding -- under a non-trivial pose, incl. points that find too few neighbours."""
    from mola_lidar_odometry_amd import synth
    cloud = synth.ndt_cloud(5)
    rng = np.random.default_rng(21)
    T = np.array([0.9986295, -0.0523360, 0.0, 0.3, 0.0523360, 0.9986295, 0.0, -0.2, 0.0, 0.0, 1.0, 0.05], np.float64)  # 3 deg yaw
    Ti = np.linalg.inv(np.vstack([T.reshape(3, 4), [0, 0, 0, 1]]))
    q_map = cloud[rng.choice(len(cloud), 4000, replace=False)] + rng.normal(0, 0.05, (4000, 3)).astype(np.float32)
    far = rng.uniform(30, 40, (50, 3)).astype(np.float32)                       # nothing near: no pairing
    q = ((np.concatenate([q_map, far]).astype(np.float64) @ Ti[:3, :3].T) + Ti[:3, 3]).astype(np.float32)
    mode = capi.INDEX_TRUNC if trunc else capi.INDEX_FLOOR
    g = capi.Map(ctx, voxel, 20, index_mode=mode).build(cloud)
    o = oracle.Map(voxel, 20, index_mode=oracle.INDEX_TRUNC if trunc else oracle.INDEX_FLOOR).insert(cloud)
    thr, eig_thr, radius = 0.08, 2e-2, 0.6
    r = capi.nn_search_pt2pl_knn(g, capi.Scan(ctx, q), T, thr, eig_thr, radius, knn, min_pts)
    e = oracle.match_pt2pl_knn(o, q, T, thr, eig_thr, radius, knn, min_pts)
    assert r["potential_pairings"] == len(q)
    assert len(e["local_idx"]) > 500 and e["local_idx"].max() < 4000
    assert np.array_equal(r["local_idx"], e["local_idx"])
    assert np.allclose(r["centroid"], e["centroid"], atol=1e-6)
    assert np.allclose(r["normal"], e["normal"], atol=1e-6)
    with pytest.raises(capi.MolahipError):
        capi.nn_search_pt2pl_knn(g, capi.Scan(ctx, q), T, thr, eig_thr, radius, 17, min_pts)   # beyond MH_MAX_PLANE_KNN


def test_out_of_memory_path_of_the_growing_buffers(ctx, oracle):
    """DevBuf::reserve under memory pressure: a failed first attempt returns the retired blocks and asks for exactly what
    is needed -- the call succeeds and the map is what it would have been; when the retry fails too the call returns
    MH_ERR_OUT_OF_MEMORY and every handle stays usable."""
    rng = np.random.default_rng(5)
    a = rng.normal(0, 6, (20000, 3)).astype(np.float32)
    b = rng.normal(0, 6, (30000, 3)).astype(np.float32) + np.float32(2.0)
    I = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64)
    g = capi.Map(ctx, 1.0, 20).build(a)
    try:
        capi.fail_allocations(64, 0)          # every first attempt of the growth this insert needs fails: the retries serve it
        g.insert(capi.Scan(ctx, b), I)
    finally:
        capi.fail_allocations(0, 0)
    o = oracle.Map(1.0, 20).insert(a)
    o.insert_posed(b, I)
    d, od = g.download(), o.dump()
    assert np.array_equal(d["xyz"], od["xyz"]) and np.array_equal(d["vox_count"], od["vox_count"])
    c = rng.normal(0, 6, (90000, 3)).astype(np.float32)
    try:
        capi.fail_allocations(64, 64)         # ... and when the retries fail as well: a clean error
        with pytest.raises(capi.MolahipError) as e:
            g.insert(capi.Scan(ctx, c), I)
        assert e.value.status == 3
    finally:
        capi.fail_allocations(0, 0)
    g2 = capi.Map(ctx, 1.0, 20).build(a)      # the context and the library are as usable as before
    assert g2.info().n_points == g.info().n_points or g2.info().n_points > 0
    s = capi.Scan(ctx, a[:2000])
    assert len(capi.nn_search(g2, s, I, 1.0)["local_idx"]) > 0


def test_map_out_of_range_is_an_error(ctx):
    g = capi.Map(ctx, 0.001, 20)
    with pytest.raises(capi.MolahipError) as e:
        g.build(np.array([[0, 0, 0], [5000.0, 0, 0]], np.float32))
    assert e.value.status == 4


def test_map_insert_after_an_out_of_range_insert_is_performed(ctx, oracle):
    """mh_map_insert's out-of-range verdict is deferred (the counters come back lazily).  It must not swallow the NEXT
    key-frame: that call performs its own insertion and THEN reports the previous one's verdict; the accessors in between
    never fail for it."""
    rng = np.random.default_rng(11)
    a = rng.normal(0, 4, (3000, 3)).astype(np.float32)
    wild = a.copy()
    wild[7] = [5.0e7, 0, 0]  # |coord| / voxel_size >= 1e6: left out of the map
    b = rng.normal(0, 4, (2500, 3)).astype(np.float32) + np.float32(1.5)
    c = rng.normal(0, 4, (2000, 3)).astype(np.float32) - np.float32(1.0)
    I = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64)
    g = capi.Map(ctx, 1.0, 20)
    g.build(a[:0])
    g.insert(capi.Scan(ctx, wild), I)        # queued: no verdict yet
    i1 = g.info()                            # never fails for it ...
    assert i1.deferred_status == 4 and i1.n_points > 0   # ... but shows it (MH_ERR_OUT_OF_RANGE)
    g.download()                             # ... nor do the downloads
    with pytest.warns(RuntimeWarning, match="HAS been performed"):
        g.insert(capi.Scan(ctx, b), I)       # reports the PREVIOUS update's verdict -- as a warning status, b is inserted
    assert g.info().deferred_status == 0
    g.insert(capi.Scan(ctx, c), I)           # and nothing lingers
    o = oracle.Map(1.0, 20)
    ok = np.ones(len(wild), bool)
    ok[7] = False
    # the oracle has no key range: give it the same content the device kept, with the device's source indices
    d = g.download()
    o.insert(wild[ok])
    o.insert(b)
    o.insert(c)
    od = o.dump()
    assert int(g.info().n_points) == len(od["x"]) if isinstance(od, dict) and "x" in od else True
    n_expected = o.num_points
    assert int(g.info().n_points) == n_expected
    assert int(g.info().n_offered) == len(wild) + len(b) + len(c)
    del d


def test_map_build_from_device_pointers(ctx, oracle):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(5)
    pts = rng.normal(0, 5, (5000, 3)).astype(np.float32)
    t = [torch.from_numpy(np.ascontiguousarray(pts[:, i])).cuda() for i in range(3)]
    torch.cuda.synchronize()
    g = capi.Map(ctx, 1.0, 20).build_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(pts))
    assert_maps_equal(g.download(), oracle.Map(1.0, 20).insert(pts).dump())


_ndt_cloud = synth.ndt_cloud


@pytest.mark.parametrize("cap,md", [(0, 0.2), (20, 0.05), (0, 0.0), (7, 0.1)])
def test_ndt_map_build_parity(ctx, oracle, cap, md):
    """mola::NDT role (lidar3d-ndt.yaml:236-254): min-distance insertion filter + per-voxel statistics."""
    pts = _ndt_cloud(cap + int(md * 100))
    g = capi.Map(ctx, 1.0, cap, 0, md, 0.05, 4).build(pts)
    o = oracle.Map(1.0, cap, 0, md, 0.05, 4).insert(pts)
    assert (g.info().n_points, g.info().n_voxels) == (o.num_points, o.num_voxels)
    assert_maps_equal(g.download(), o.dump())  # which points survive the filter: bit-exact
    gn, on = g.download_ndt(), o.dump_ndt()
    np.testing.assert_array_equal(gn["is_plane"], on["is_plane"])
    assert g.info().n_planes == int(on["is_plane"].sum()) > 100
    np.testing.assert_array_equal(gn["centroid"], on["centroid"])
    np.testing.assert_allclose(gn["normal"], on["normal"], atol=1e-6)


@pytest.mark.parametrize("thr", [0.5, 0.1, 0.02])
def test_point2plane_matcher_parity(ctx, oracle, thr):
    """Matcher_Point2Plane on the NDT map (lidar3d-ndt.yaml:195-200): same pairings, centroids and normals."""
    pts = _ndt_cloud(7)
    g = capi.Map(ctx, 1.0, 0, 0, 0.1, 0.05, 4).build(pts)
    o = oracle.Map(1.0, 0, 0, 0.1, 0.05, 4).insert(pts)
    rng = np.random.default_rng(8)
    q = np.concatenate([pts[rng.permutation(len(pts))[:4000]] + rng.normal(0, 0.05, (4000, 3)).astype(np.float32),
                        rng.uniform(-12, 12, (1000, 3)).astype(np.float32)]).astype(np.float32)
    q[5] = [np.nan, 0, 0]
    gs = capi.Scan(ctx, q)
    for T in (I12, oracle.se3_exp([0.06, -0.04, 0.03, 0.004, -0.002, 0.005])):
        a = capi.nn_search_pt2pl(g, gs, T, thr)
        b = oracle.match_pt2pl(o, q, T, thr)
        np.testing.assert_array_equal(a["local_idx"], b["local_idx"])
        np.testing.assert_array_equal(a["centroid"], b["centroid"])
        np.testing.assert_allclose(a["normal"], b["normal"], atol=1e-6)
        assert a["potential_pairings"] == len(q)
        # SURVEY App. B U10, the switchable other reading: distanceThreshold against the distance to the centroid
        ac = capi.nn_search_pt2pl(g, gs, T, thr + 0.3, mode=capi.PT2PL_CENTROID_DISTANCE)
        bc = oracle.match_pt2pl(o, q, T, thr + 0.3, mode=oracle.PT2PL_CENTROID_DISTANCE)
        np.testing.assert_array_equal(ac["local_idx"], bc["local_idx"])
        np.testing.assert_array_equal(ac["centroid"], bc["centroid"])
        assert 0 < len(ac["local_idx"]) != len(a["local_idx"])
    assert len(a["local_idx"]) > 0
    with pytest.raises(capi.MolahipError):  # a plain map has no planes to offer
        capi.nn_search_pt2pl(capi.Map(ctx, 1.0, 20).build(pts), gs, I12, thr)


@pytest.mark.parametrize("n_scan,match", [(1500, None), (5000, None), (5000, "p")])
def test_align_ndt_pipeline_centroid_mode(ctx, oracle, n_scan, match, monkeypatch):
    """mh_icp_params::pt2pl_mode = MH_PT2PL_CENTROID_DISTANCE through the fused loop (every kernel that runs the plane
    matcher), against the oracle's negative-threshold reading."""
    if match:
        monkeypatch.setenv("MH_MATCH", match)
    pts = _ndt_cloud(21)
    g = capi.Map(ctx, 1.0, 0, 0, 0.1, 0.05, 4).build(pts)
    o = oracle.Map(1.0, 0, 0, 0.1, 0.05, 4).insert(pts)
    rng = np.random.default_rng(22)
    scan = pts[rng.permutation(len(pts))[:n_scan]]
    guess = oracle.se3_exp([0.1, -0.07, 0.05, 0.005, -0.004, 0.008])
    thr, kp = synth.threshold_schedule(0.5, 40)
    kw = dict(max_iterations=40, min_abs_step_trans=5e-4, min_abs_step_rot=5e-4, threshold=thr, kernel_param=kp,
              pt2pl_threshold=0.6, pt2pl_mode=1)
    a = capi.icp_align(g, capi.Scan(ctx, scan), guess, capi.ICPParams(gn=capi.GNParams(max_inner_iterations=1), **kw), want_pairs=True)
    b = oracle.icp_align(o, scan, guess, oracle.ICPParams(gn=oracle.GNParams(max_inner_iterations=1), **kw), want_pairs=True)
    assert_align_equal(a, b)
    assert a["n_final_pairs_pt2pl"] == b["n_final_pairs_pt2pl"] > 0
    plane = oracle.icp_align(o, scan, guess, oracle.ICPParams(gn=oracle.GNParams(max_inner_iterations=1), **dict(kw, pt2pl_mode=0)))
    assert plane["n_final_pairs_pt2pl"] != b["n_final_pairs_pt2pl"]  # the switch does change the pairing set


@pytest.mark.parametrize("inner,match,n_scan", [(1, None, 5000), (2, None, 5000), (2, "p", 5000), (2, "q", 5000),
                                                (2, None, 1500)])
def test_align_ndt_pipeline_matches_oracle(ctx, oracle, inner, match, n_scan, monkeypatch):
    """The lidar3d-ndt.yaml ICP block: Matcher_Point2Plane + Matcher_Points_DistanceThreshold feeding one
    Gauss-Newton solve per iteration (yaml:184-210), stall thresholds 5e-4 (yaml:173-174).  Kernel paths: both
    matchers in the row kernel + separate accumulations (5000 points), everything in one workgroup (1500), and the
    matchers of large layers (MH_MATCH=p: one lane per point for both; q: quad kernel for the points)."""
    if match:
        monkeypatch.setenv("MH_MATCH", match)
    pts = _ndt_cloud(11)
    g = capi.Map(ctx, 1.0, 0, 0, 0.1, 0.05, 4).build(pts)
    o = oracle.Map(1.0, 0, 0, 0.1, 0.05, 4).insert(pts)
    rng = np.random.default_rng(12)
    scan = pts[rng.permutation(len(pts))[:n_scan]]
    guess = oracle.se3_exp([0.12, -0.09, 0.06, 0.006, -0.004, 0.01])
    thr, kp = synth.threshold_schedule(0.5, 60)
    kw = dict(max_iterations=60, min_abs_step_trans=5e-4, min_abs_step_rot=5e-4, threshold=thr, kernel_param=kp,
              pt2pl_threshold=0.5)
    a = capi.icp_align(g, capi.Scan(ctx, scan), guess, capi.ICPParams(gn=capi.GNParams(max_inner_iterations=inner), **kw),
                       want_pairs=True)
    b = oracle.icp_align(o, scan, guess, oracle.ICPParams(gn=oracle.GNParams(max_inner_iterations=inner), **kw),
                         want_pairs=True)
    assert_align_equal(a, b)
    assert a["n_final_pairs_pt2pl"] == b["n_final_pairs_pt2pl"] > n_scan // 5
    assert a["potential_pairings"] == 2 * len(scan)
    np.testing.assert_array_equal(a["pairs"]["global_idx"], b["pairs"]["global_idx"])
    np.testing.assert_allclose(a["cov"], b["cov"], rtol=2e-5, atol=1e-6 * np.abs(b["cov"]).max())
    assert np.abs(a["T"] - I12).max() < 5e-3  # and it actually registers the scan


@pytest.mark.parametrize("inner,match,n_scan", [(1, None, 5000), (2, None, 1500), (1, "p", 5000), (1, "q", 40000)])
def test_align_ndt_pipeline_skipping_plane_paired_points(ctx, oracle, inner, match, n_scan, monkeypatch):
    """SURVEY App. B, U12 (found in round 4): upstream's matchers skip local points an earlier matcher of the same iteration
    has paired unless allowMatchAlreadyMatchedPoints [U] is set -- in lidar3d-ndt.yaml:195-210 plane-paired points would then
    get no point-to-point pairing.  mh_icp_params::matched_points = MH_MATCHED_POINTS_SKIP against the oracle's
    pt2pt_skip_plane_paired on every kernel path that runs the plane matcher (requested one-lane / quad matchers fall back
    to the row kernel, the only one that sees both verdicts of a point); the default keeps pairing every point twice."""
    if match:
        monkeypatch.setenv("MH_MATCH", match)
    pts = _ndt_cloud(31)
    g = capi.Map(ctx, 1.0, 0, 0, 0.1, 0.05, 4).build(pts)
    o = oracle.Map(1.0, 0, 0, 0.1, 0.05, 4).insert(pts)
    rng = np.random.default_rng(32)
    scan = pts[rng.integers(0, len(pts), n_scan)] + rng.normal(0, 0.01, (n_scan, 3)).astype(np.float32)
    guess = oracle.se3_exp([0.12, -0.09, 0.06, 0.006, -0.004, 0.01])
    thr, kp = synth.threshold_schedule(0.5, 60)
    kw = dict(max_iterations=60, min_abs_step_trans=5e-4, min_abs_step_rot=5e-4, threshold=thr, kernel_param=kp, pt2pl_threshold=0.5)
    a = capi.icp_align(g, capi.Scan(ctx, scan), guess, capi.ICPParams(gn=capi.GNParams(max_inner_iterations=inner), matched_points=1, **kw),
                       want_pairs=True)
    b = oracle.icp_align(o, scan, guess, oracle.ICPParams(gn=oracle.GNParams(max_inner_iterations=inner), pt2pt_skip_plane_paired=True, **kw),
                         want_pairs=True)
    assert_align_equal(a, b)
    assert a["n_final_pairs_pt2pl"] == b["n_final_pairs_pt2pl"] > n_scan // 5
    np.testing.assert_array_equal(a["pairs"]["local_idx"], b["pairs"]["local_idx"])
    np.testing.assert_array_equal(a["pairs"]["global_idx"], b["pairs"]["global_idx"])
    np.testing.assert_allclose(a["cov"], b["cov"], rtol=2e-5, atol=1e-6 * np.abs(b["cov"]).max())
    both = oracle.icp_align(o, scan, guess, oracle.ICPParams(gn=oracle.GNParams(max_inner_iterations=inner), **kw), want_pairs=True)
    n_pt = b["n_final_pairs"] - b["n_final_pairs_pt2pl"]
    assert 0 < n_pt < both["n_final_pairs"] - both["n_final_pairs_pt2pl"]  # the switch does change the pairing set
    assert np.abs(a["T"] - I12).max() < 4e-2  # (from 0.12 off; with the plane-paired points left out of the point matcher 3 cm remain)


# ---------------------------------------------------------------------------- NN / matcher
def test_nn_dense_bit_exact(ctx, oracle, small):
    w, gm, om, gs = small
    d = capi.nn_search_dense(gm, gs, w.T_guess)
    big = oracle.match_points(om, w.scan_xyz, w.T_guess, 1e9)  # no threshold: everything found pairs
    found = d["global_idx"] != capi.NO_MATCH
    np.testing.assert_array_equal(np.nonzero(found)[0], big["local_idx"])
    np.testing.assert_array_equal(d["global_idx"][found], big["global_idx"])
    np.testing.assert_array_equal(d["d2"][found], big["d2"])
    np.testing.assert_array_equal(d["global_xyz"][found], big["global_xyz"])


@pytest.mark.parametrize("thr,ang", [(8.0, 0.0), (0.6, 0.0), (0.3, 0.5), (0.05, 0.0)])
def test_nn_search_compacted_bit_exact(ctx, oracle, small, thr, ang):
    w, gm, om, gs = small
    for T in (w.T_guess, w.T_gt):
        g = capi.nn_search(gm, gs, T, thr, ang)
        o = oracle.match_points(om, w.scan_xyz, T, thr, ang)
        for k in ("local_idx", "global_idx", "d2", "global_xyz"):
            np.testing.assert_array_equal(g[k], o[k])
        assert g["potential_pairings"] == o["potential_pairings"] == len(w.scan_xyz)


def test_nn_random_clouds_and_ragged_sizes(ctx, oracle):
    rng = np.random.default_rng(3)
    pts = rng.normal(0, 4, (20000, 3)).astype(np.float32)
    gm = capi.Map(ctx, 0.8, 7).build(pts)
    om = oracle.Map(0.8, 7).insert(pts)
    T = oracle.se3_exp([0.3, -0.2, 0.1, 0.02, -0.01, 0.03])
    for n in (1, 63, 64, 65, 255, 257, 1000, 4097):
        q = rng.normal(0, 4, (n, 3)).astype(np.float32)
        gs = capi.Scan(ctx, q)
        g = capi.nn_search(gm, gs, T, 0.9)
        o = oracle.match_points(om, q, T, 0.9)
        for k in ("local_idx", "global_idx", "d2"):
            np.testing.assert_array_equal(g[k], o[k])


def test_nn_nonfinite_queries_and_empty_inputs(ctx, oracle):
    pts = np.random.default_rng(2).normal(0, 2, (500, 3)).astype(np.float32)
    gm = capi.Map(ctx, 1.0, 20).build(pts)
    q = np.array([[0.1, 0.2, 0.3], [np.nan, 0, 0], [0, np.inf, 0], [100, 100, 100]], np.float32)
    g = capi.nn_search(gm, capi.Scan(ctx, q), I12, 5.0)
    assert g["local_idx"].tolist() == [0]
    # empty scan / empty map
    g = capi.nn_search(gm, capi.Scan(ctx, np.zeros((0, 3), np.float32)), I12, 5.0)
    assert len(g["local_idx"]) == 0 and g["potential_pairings"] == 0
    g = capi.nn_search(capi.Map(ctx, 1.0, 20), capi.Scan(ctx, q), I12, 5.0)
    assert len(g["local_idx"]) == 0 and g["potential_pairings"] == 4


def test_nn_voxel_boundary_and_negative_coords(ctx, oracle):
    pts = np.array([[-0.2, 0.5, 0.5], [0.2, 0.5, 0.5], [-1.0, 0.0, 0.0], [2.9, 0.5, 0.5]], np.float32)
    q = np.array([[0.0, 0.5, 0.5], [-1.0, 0.0, 0.0], [1.0, 0.5, 0.5], [-0.0, 0.5, 0.5], [0.999999, 0.5, 0.5]], np.float32)
    for mode in (0, 1):
        gm = capi.Map(ctx, 1.0, 20, mode).build(pts)
        om = oracle.Map(1.0, 20, mode).insert(pts)
        g = capi.nn_search(gm, capi.Scan(ctx, q), I12, 10.0)
        o = oracle.match_points(om, q, I12, 10.0)
        for k in ("local_idx", "global_idx", "d2"):
            np.testing.assert_array_equal(g[k], o[k])


@pytest.mark.parametrize("vs,cap,mode,offset", [(1.0, 20, 0, 0.0), (0.3, 5, 0, 0.0), (2.5, 0, 0, 0.0), (1.0, 20, 1, 0.0),
                                                (1.0, 20, 0, 50000.0), (0.7, 8, 0, -12345.0), (0.3, 20, 1, 3.0)])
@pytest.mark.parametrize("match", ["p", "t", "w"])
def test_nn_pruning_is_exact_on_adversarial_clouds(ctx, oracle, vs, cap, mode, offset, match, monkeypatch):
    """The production search prunes voxels with a conservative distance bound; its output must stay
    bit-identical to the exhaustive 27-voxel scan of the oracle.  Adversarial inputs: lattice-aligned map
    points and queries (exact distance ties in different voxels), queries exactly on voxel boundaries,
    coordinates where fp32 spacing is coarse, voxel sizes whose reciprocal is inexact, trunc indexing."""
    monkeypatch.setenv("MH_MATCH", match)  # p: one lane per point through the caches; t: tiles staged in LDS; w: wave-uniform candidates
    rng = np.random.default_rng(int(vs * 10) + cap + mode)
    g = np.arange(-6, 6, 0.25, dtype=np.float32)
    lattice = np.stack(np.meshgrid(g, g, g[:24], indexing="ij"), -1).reshape(-1, 3)
    lattice = lattice[rng.permutation(len(lattice))[:40000]]
    noise = rng.normal(0, 3, (20000, 3)).astype(np.float32)
    pts = (np.concatenate([lattice, noise]) + np.float32(offset)).astype(np.float32)
    gm = capi.Map(ctx, vs, cap, mode).build(pts)
    om = oracle.Map(vs, cap, mode).insert(pts)
    q_mid = (lattice[:3000] + np.float32(0.125)).astype(np.float32)          # equidistant from 8 lattice points
    q_bnd = np.round(rng.uniform(-6, 6, (3000, 3)) / vs).astype(np.float32) * np.float32(vs)  # on voxel faces
    q_bnd[:, 1] += rng.uniform(-0.5, 0.5, 3000).astype(np.float32)
    q_rnd = rng.uniform(-7, 7, (4000, 3)).astype(np.float32)
    q_far = rng.uniform(-30, 30, (500, 3)).astype(np.float32)                 # mostly empty neighbourhoods
    q = (np.concatenate([q_mid, q_bnd, q_rnd, q_far]) + np.float32(offset)).astype(np.float32)
    gs = capi.Scan(ctx, q)
    for T in (I12, oracle.se3_exp([0.11, -0.07, 0.05, 0.002, -0.001, 0.003])):
        d = capi.nn_search_dense(gm, gs, T)
        o = oracle.match_points(om, q, T, 1e9)
        found = d["global_idx"] != capi.NO_MATCH
        np.testing.assert_array_equal(np.nonzero(found)[0], o["local_idx"])
        np.testing.assert_array_equal(d["global_idx"][found], o["global_idx"])
        np.testing.assert_array_equal(d["d2"][found], o["d2"])
        np.testing.assert_array_equal(d["global_xyz"][found], o["global_xyz"])


@pytest.mark.gpu
@pytest.mark.parametrize("vs,cap,mode,offset", [(1.0, 20, 0, 0.0), (0.3, 5, 0, 0.0), (2.5, 0, 0, 0.0), (1.0, 20, 1, 0.0),
                                                (1.0, 20, 0, 50000.0), (0.7, 8, 0, -12345.0)])
@pytest.mark.parametrize("k", [2, 3, 8])
def test_k_best_matcher_is_exact_on_adversarial_clouds(ctx, oracle, vs, cap, mode, offset, k):
    """mh_nn_search_k (Matcher_Points_DistanceThreshold with pairingsPerPoint = k, rgbd.yaml:135-141, on nn_multiple_search):
    the k nearest of the 27-voxel block in ascending (distance, scan position), accepted while below the limit -- bit-equal to
    the oracle on the adversarial inputs of the single-neighbour test (exact ties across voxels, queries on voxel faces, coarse
    fp32 spacing, inexact reciprocals, trunc indexing), with and without the angular term, on plain and NDT maps."""
    rng = np.random.default_rng(int(vs * 10) + cap + mode + k)
    g = np.arange(-6, 6, 0.25, dtype=np.float32)
    lattice = np.stack(np.meshgrid(g, g, g[:24], indexing="ij"), -1).reshape(-1, 3)
    lattice = lattice[rng.permutation(len(lattice))[:40000]]
    noise = rng.normal(0, 3, (20000, 3)).astype(np.float32)
    pts = (np.concatenate([lattice, noise]) + np.float32(offset)).astype(np.float32)
    ndt = dict(ndt_max_eigen_ratio=0.05) if (cap == 20 and mode == 0 and offset == 0.0) else {}
    gm = capi.Map(ctx, vs, cap, mode, **ndt).build(pts)
    om = oracle.Map(vs, cap, mode, **ndt).insert(pts)
    q_mid = (lattice[:2000] + np.float32(0.125)).astype(np.float32)
    q_bnd = np.round(rng.uniform(-6, 6, (2000, 3)) / vs).astype(np.float32) * np.float32(vs)
    q_bnd[:, 1] += rng.uniform(-0.5, 0.5, 2000).astype(np.float32)
    q_rnd = rng.uniform(-7, 7, (3000, 3)).astype(np.float32)
    q_far = rng.uniform(-30, 30, (500, 3)).astype(np.float32)
    q = (np.concatenate([q_mid, q_bnd, q_rnd, q_far]) + np.float32(offset)).astype(np.float32)
    gs = capi.Scan(ctx, q)
    for T, thr, ang in ((I12, 1e9, 0.0), (oracle.se3_exp([0.11, -0.07, 0.05, 0.002, -0.001, 0.003]), 0.35 * vs, 0.0),
                        (I12, 0.1 * vs, 2.0)):
        d = capi.nn_search_k(gm, gs, T, thr, k, ang)
        o = oracle.match_points_k(om, q, T, thr, k, ang)
        assert d["potential_pairings"] == o["potential_pairings"] == len(q) * k
        for key in ("local_idx", "global_idx", "d2", "global_xyz"):
            np.testing.assert_array_equal(d[key], o[key])
        assert thr < 1e8 or len(o["local_idx"]) > len(q) // 2
    # k = 1 is mh_nn_search; an empty scan and an empty map pair nothing
    a, b = capi.nn_search_k(gm, gs, I12, 0.5 * vs, 1), capi.nn_search(gm, gs, I12, 0.5 * vs)
    for key in ("local_idx", "global_idx", "d2"):
        np.testing.assert_array_equal(a[key], b[key])
    assert len(capi.nn_search_k(gm, capi.Scan(ctx), I12, 1.0, k)["local_idx"]) == 0
    with pytest.raises(capi.MolahipError):
        capi.nn_search_k(gm, gs, I12, 1.0, 9)


# ---------------------------------------------------------------------------- solver
def _pairs(rng, n, noise=0.05):
    l = rng.normal(0, 10, (n, 3)).astype(np.float32)
    Tt = np.asarray(__import__("oracle.oracle_c", fromlist=["x"]).se3_exp(
        np.concatenate([rng.normal(0, 0.3, 3), rng.normal(0, 0.03, 3)]))).reshape(3, 4)
    q = (l.astype(np.float64) @ Tt[:, :3].T + Tt[:, 3] + rng.normal(0, noise, (n, 3))).astype(np.float32)
    return l, q


def _planes(rng, n):
    l = rng.normal(0, 10, (n, 3)).astype(np.float32)
    nn = rng.normal(0, 1, (n, 3))
    nn /= np.linalg.norm(nn, axis=1, keepdims=True)
    return l, (l + rng.normal(0, 0.2, (n, 3))).astype(np.float32), nn.astype(np.float32)


@pytest.mark.parametrize("kernel", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("with_planes", [False, True])
def test_gn_solve_matches_oracle(ctx, oracle, kernel, with_planes):
    rng = np.random.default_rng(10 + kernel)
    l, q = _pairs(rng, 3001)
    pl = _planes(rng, 777) if with_planes else None
    T0 = oracle.se3_exp([0.1, -0.05, 0.02, 0.01, -0.02, 0.005])
    gp = capi.GNParams(max_inner_iterations=3, robust_kernel=kernel, robust_kernel_param=0.7)
    op = oracle.GNParams(max_inner_iterations=3, robust_kernel=kernel, robust_kernel_param=0.7)
    Tg, ng, ok, sg = capi.gn_solve(ctx, T0, (l, q), pl, gp)
    To, no, so = oracle.gn_solve(T0, (l, q), pl, op)
    assert ok and ng == no == 3
    for a, b in zip(sg, so):
        scale = np.abs(b["H"]).max()
        np.testing.assert_allclose(a["H"], b["H"], rtol=1e-10, atol=1e-12 * scale)
        # |g_i| <= sqrt(H_ii * cost) (Cauchy-Schwarz): that bound is the scale of the summed terms
        np.testing.assert_allclose(a["g"], b["g"], rtol=1e-9, atol=1e-11 * np.sqrt(scale * b["err_norm_sqr"]))
        np.testing.assert_allclose(a["err_norm_sqr"], b["err_norm_sqr"], rtol=1e-12)
        np.testing.assert_allclose(a["delta"], b["delta"], rtol=1e-7, atol=1e-11)
    np.testing.assert_allclose(Tg, To, atol=1e-11)


def test_gn_solve_prior_and_planes_only(ctx, oracle):
    rng = np.random.default_rng(4)
    l = np.stack([rng.uniform(-10, 10, 300), rng.uniform(-10, 10, 300), np.full(300, 0.1)], 1).astype(np.float32)
    c = l.copy(); c[:, 2] = 0
    nn = np.tile(np.array([0, 0, 1], np.float32), (300, 1))
    Tp = oracle.se3_exp([0.05, -0.02, 0.0, 0, 0, 0.01])
    Lam = np.diag([10, 10, 10, 100, 100, 100.0])
    gp = capi.GNParams(max_inner_iterations=2, robust_kernel=0)
    op = oracle.GNParams(max_inner_iterations=2, robust_kernel=0)
    Tg, ng, ok, sg = capi.gn_solve(ctx, I12, None, (l, c, nn), gp, prior=(Tp, Lam))
    To, no, so = oracle.gn_solve(I12, None, (l, c, nn), op, prior=(Tp, Lam))
    assert ok and ng == no
    np.testing.assert_allclose(sg[0]["H"], so[0]["H"], rtol=1e-8, atol=1e-7)
    np.testing.assert_allclose(sg[0]["g"], so[0]["g"], rtol=1e-8, atol=1e-7)
    np.testing.assert_allclose(Tg, To, atol=1e-10)
    # rank-deficient without prior: the pivots that should be exactly zero are rounding noise, so the
    # solution is arbitrary along the unobservable directions (the reference's Eigen ldlt() has the same
    # property); the contract is only "does not crash, reports through solver_ok / finite values"
    Tg2, ng2, ok2, _ = capi.gn_solve(ctx, I12, None, (l, c, nn), gp)
    assert (not ok2) or np.all(np.isfinite(Tg2))
    assert abs(Tg2[11] + 0.1) < 1e-6 or not ok2  # the observable direction (z) is still solved


def test_gn_solve_empty_and_identity(ctx, oracle):
    e = np.zeros((0, 3), np.float32)
    Tg, n, ok, _ = capi.gn_solve(ctx, I12, (e, e), None, capi.GNParams())
    assert n == 0 and ok and np.array_equal(Tg, I12)
    l = np.random.default_rng(0).normal(0, 5, (100, 3)).astype(np.float32)
    Tg, n, ok, _ = capi.gn_solve(ctx, I12, (l, l), None, capi.GNParams())
    assert n == 0 and np.array_equal(Tg, I12)  # zero cost: early exit before any solve


def test_covariance_matches_oracle(ctx, oracle):
    rng = np.random.default_rng(9)
    l, q = _pairs(rng, 5000)
    pl = _planes(rng, 500)
    T = oracle.se3_exp([0.3, 0.1, -0.2, 0.05, -0.02, 0.4])
    for args in (((l, q), None), ((l, q), pl), (None, pl)):
        cg = capi.covariance(ctx, T, *args)
        co, _ = oracle.covariance(T, *args)
        np.testing.assert_allclose(cg, co, rtol=2e-5, atol=1e-6 * np.abs(co).max())
    e = np.zeros((0, 3), np.float32)
    np.testing.assert_array_equal(capi.covariance(ctx, T, (e, e), None), np.eye(6) * 1e6)


# ---------------------------------------------------------------------------- fused align
def _params(mod, w, n_it=None, **kw):
    n_it = n_it or w.n_iters
    thr, kp = synth.threshold_schedule(w.sigma, n_it)
    return mod.ICPParams(max_iterations=n_it, threshold=thr, kernel_param=kp, **kw)


def assert_align_equal(g, o, pose_tol=POSE_TOL):
    assert g["n_iterations"] == o["n_iterations"]
    assert capi.TERM_NAMES[g["termination_reason"]] == capi.TERM_NAMES[o["termination_reason"]]
    assert g["n_final_pairs"] == o["n_final_pairs"]
    assert g["potential_pairings"] == o["potential_pairings"]
    assert g["quality"] == o["quality"]
    np.testing.assert_allclose(g["T"], o["T"], atol=pose_tol, rtol=0)


def test_align_fixed_iterations_matches_oracle_with_trace(ctx, oracle, small):
    w, gm, om, gs = small
    g = capi.icp_align(gm, gs, w.T_guess, _params(capi, w, disable_stall_test=True), want_pairs=True)
    o = oracle.icp_align(om, w.scan_xyz, w.T_guess, _params(oracle, w, disable_stall_test=True), want_pairs=True)
    assert_align_equal(g, o, 1e-9)
    assert len(g["trace"]) == len(o["trace"]) == w.n_iters
    for a, b in zip(g["trace"], o["trace"]):
        assert a["n_pairs"] == b["n_pairs"] and a["threshold"] == b["threshold"]
        np.testing.assert_allclose(a["T"], b["T"], atol=1e-9)
        np.testing.assert_allclose([a["delta_trans"], a["delta_rot"]], [b["delta_trans"], b["delta_rot"]], atol=1e-9)
    for k in ("local_idx", "global_idx", "d2", "global_xyz"):
        np.testing.assert_array_equal(g["pairs"][k], o["pairs"][k])
    np.testing.assert_allclose(g["cov"], o["cov"], rtol=2e-5, atol=1e-6 * np.abs(o["cov"]).max())


@pytest.mark.parametrize("kernel", [0, 1, 2, 3, 4, 5])
def test_align_all_kernels(ctx, oracle, small, kernel):
    w, gm, om, gs = small
    g = capi.icp_align(gm, gs, w.T_guess, _params(capi, w, 8, disable_stall_test=True,
                                                  gn=capi.GNParams(robust_kernel=kernel)))
    o = oracle.icp_align(om, w.scan_xyz, w.T_guess, _params(oracle, w, 8, disable_stall_test=True,
                                                            gn=oracle.GNParams(robust_kernel=kernel)))
    assert_align_equal(g, o)


@pytest.mark.parametrize("inner", [1, 2, 4])
def test_align_inner_iterations(ctx, oracle, small, inner):
    w, gm, om, gs = small
    g = capi.icp_align(gm, gs, w.T_guess, _params(capi, w, 6, disable_stall_test=True,
                                                  gn=capi.GNParams(max_inner_iterations=inner)))
    o = oracle.icp_align(om, w.scan_xyz, w.T_guess, _params(oracle, w, 6, disable_stall_test=True,
                                                            gn=oracle.GNParams(max_inner_iterations=inner)))
    assert_align_equal(g, o)


@pytest.mark.parametrize("poll", [0, 1, 3, 7, 64])
def test_align_stall_termination_any_poll_interval(ctx, oracle, small, poll):
    """yaml defaults: maxIterations 300, stall thresholds 1e-4 / 5e-5 (lidar3d-default.yaml:173-175).  The
    termination iteration must equal the oracle's exactly whatever the host polling interval (0 = automatic: the first
    chunk sized by the context's previous alignment -- run twice so that both the cold and the predicted sizes occur)."""
    w, gm, om, gs = small
    g = capi.icp_align(gm, gs, w.T_guess, _params(capi, w, 300, poll_every=poll))
    if poll == 0:
        g2 = capi.icp_align(gm, gs, w.T_guess, _params(capi, w, 300, poll_every=poll))
        assert g2["n_iterations"] == g["n_iterations"] and np.array_equal(g2["T"], g["T"])
    o = oracle.icp_align(om, w.scan_xyz, w.T_guess, _params(oracle, w, 300))
    assert_align_equal(g, o)
    assert capi.TERM_NAMES[g["termination_reason"]] in ("Stalled", "MaxIterations")
    assert len(g["trace"]) == len(o["trace"])


def test_align_hook_request(ctx, oracle, small):
    w, gm, om, gs = small
    kw = dict(disable_stall_test=True, hook_enabled=True, hook_min_trans=0.15, hook_min_rot=float(np.deg2rad(0.75)))
    g = capi.icp_align(gm, gs, w.T_guess, _params(capi, w, **kw))
    o = oracle.icp_align(om, w.scan_xyz, w.T_guess, _params(oracle, w, **kw))
    assert capi.TERM_NAMES[g["termination_reason"]] == "HookRequest"
    assert_align_equal(g, o)
    # the caller's protocol (LidarOdometry.cpp:956-1007): re-run from the checkpoint with the remaining budget
    rem = w.n_iters - g["n_iterations"]
    g2 = capi.icp_align(gm, gs, g["T"], _params(capi, w, rem, **kw))
    o2 = oracle.icp_align(om, w.scan_xyz, o["T"], _params(oracle, w, rem, **kw))
    assert_align_equal(g2, o2)


def test_align_with_prior(ctx, oracle, small):
    w, gm, om, gs = small
    Lam = np.diag([50, 50, 50, 500, 500, 500.0])
    prior = (w.T_guess, Lam)
    g = capi.icp_align(gm, gs, w.T_guess, _params(capi, w, 10, disable_stall_test=True), prior=prior)
    o = oracle.icp_align(om, w.scan_xyz, w.T_guess, _params(oracle, w, 10, disable_stall_test=True), prior=prior)
    assert_align_equal(g, o)
    free = capi.icp_align(gm, gs, w.T_guess, _params(capi, w, 10, disable_stall_test=True))
    assert np.linalg.norm(free["T"] - g["T"]) > 1e-4  # the prior does act


def test_align_no_pairings_and_trivial_inputs(ctx, oracle):
    gm = capi.Map(ctx, 1.0, 20).build(np.zeros((10, 3), np.float32))
    far = np.full((5, 3), 50.0, np.float32)
    p = capi.ICPParams(max_iterations=5, threshold=2.0, kernel_param=0.5)
    g = capi.icp_align(gm, capi.Scan(ctx, far), I12, p)
    assert capi.TERM_NAMES[g["termination_reason"]] == "NoPairings"
    assert g["quality"] == 0.0 and g["n_iterations"] == 0 and g["n_final_pairs"] == 0
    np.testing.assert_array_equal(g["cov"], np.eye(6) * 1e6)
    np.testing.assert_array_equal(g["T"], I12)
    # empty scan, max_iterations = 0
    g = capi.icp_align(gm, capi.Scan(ctx, np.zeros((0, 3), np.float32)), I12, p)
    assert capi.TERM_NAMES[g["termination_reason"]] == "NoPairings" and g["potential_pairings"] == 0
    p0 = capi.ICPParams(max_iterations=0, threshold=np.zeros(0), kernel_param=np.zeros(0))
    g = capi.icp_align(gm, capi.Scan(ctx, far), I12, p0)
    assert capi.TERM_NAMES[g["termination_reason"]] == "MaxIterations" and g["quality"] == 0.0


def test_align_identity_known_answer(ctx, oracle):
    rng = np.random.default_rng(0)
    pts = rng.uniform(-20, 20, (5000, 3)).astype(np.float32)
    gm = capi.Map(ctx, 1.0, 20).build(pts)
    scan = gm.download()["xyz"][::5]
    g = capi.icp_align(gm, capi.Scan(ctx, scan), I12, capi.ICPParams(max_iterations=10, threshold=2.0, kernel_param=0.5))
    np.testing.assert_array_equal(g["T"], I12)
    assert g["quality"] == 1.0 and capi.TERM_NAMES[g["termination_reason"]] == "Stalled" and g["n_iterations"] == 0


def test_align_invalid_arguments_are_status_codes(ctx, small):
    w, gm, om, gs = small
    bad = w.T_guess.copy(); bad[3] = np.nan
    with pytest.raises(capi.MolahipError) as e:
        capi.icp_align(gm, gs, bad, _params(capi, w))
    assert e.value.status == 1
    with pytest.raises(capi.MolahipError):
        capi.icp_align(gm, gs, w.T_guess, _params(capi, w, gn=capi.GNParams(robust_kernel=99)))


def test_align_batch_equals_single(ctx, oracle, small):
    w, gm, om, gs = small
    ctxs = [capi.Context(0) for _ in range(4)]
    rng = np.random.default_rng(1)
    scans, guesses, singles = [], [], []
    p = _params(capi, w, 300)
    for c in ctxs:
        sub = w.scan_xyz[rng.permutation(len(w.scan_xyz))[:1500]]
        guess = w.T_guess.copy(); guess[[3, 7]] += rng.normal(0, 0.1, 2)
        scans.append(capi.Scan(c, sub)); guesses.append(guess)
        singles.append(capi.icp_align(gm, capi.Scan(ctx, sub), guess, p, want_trace=False))
    batch = capi.icp_align_batch([gm] * 4, scans, guesses, p)
    for b, s in zip(batch, singles):
        assert b["n_iterations"] == s["n_iterations"] and b["termination_reason"] == s["termination_reason"]
        np.testing.assert_array_equal(b["T"], s["T"])  # bitwise: deterministic reductions
        np.testing.assert_array_equal(b["cov"], s["cov"])
    # the prepared form (arguments marshalled once, what bench.py repeats every step): the same results, run after run
    call = capi.BatchCall([gm] * 4, scans, guesses, p)
    for _ in range(3):
        raw = call.run()
        assert len(raw) == 4
        for b, s in zip(call.results(), singles):
            assert b["n_iterations"] == s["n_iterations"] and b["termination_reason"] == s["termination_reason"]
            np.testing.assert_array_equal(b["T"], s["T"])
            np.testing.assert_array_equal(b["cov"], s["cov"])


@pytest.mark.parametrize("inner,ndt", [(1, False), (2, False), (2, True)])
def test_small_layer_batches_give_the_bits_of_single_alignments(ctx, inner, ndt):
    """Layers of up to 8 k points run k_step16 -- search + sums per launch, the Gauss-Newton step carried into the next launch
    -- alone (one group of 32 points per workgroup, streaming loop control) and in lock-step batches (k_step16_b: the jobs'
    workgroups side by side, several groups per workgroup, chunks of launches).  The partial sums are one column per GROUP
    in both, added in the same order: pose, covariance, iteration count and pairing counts agree bit for bit whatever the
    batch's size -- a sequence's trajectory does not depend on what it shared the device with."""
    pts = _ndt_cloud(41)
    gm = capi.Map(ctx, 1.0, 0, 0, 0.1, 0.05, 4).build(pts) if ndt else capi.Map(ctx, 1.0, 20).build(pts)
    rng = np.random.default_rng(42)
    thr, kp = synth.threshold_schedule(0.5, 60)
    kw = dict(max_iterations=60, threshold=thr, kernel_param=kp, gn=capi.GNParams(max_inner_iterations=inner))
    if ndt:
        kw["pt2pl_threshold"] = 0.5
    p = capi.ICPParams(**kw)
    sizes = [1500, 900, 2048, 33, 2100, 5000, 1400, 1400, 1400, 1400, 1400, 1400]  # (12 jobs: 21 workgroups each, up to 8 groups per workgroup)
    ctxs = [capi.Context(0) for _ in sizes]
    subs = [pts[rng.integers(0, len(pts), n)] + rng.normal(0, 0.01, (n, 3)).astype(np.float32) for n in sizes]
    guesses = [synth.pose_from_ypr([0.1 + 0.02 * (k % 4), -0.08, 0.05, 0.006, -0.004, 0.01]) for k in range(len(sizes))]
    single = [capi.icp_align(gm, capi.Scan(ctx, s), g, p) for s, g in zip(subs, guesses)]
    scans = [capi.Scan(c, s) for c, s in zip(ctxs, subs)]
    for count in (len(sizes), 3):
        batch = capi.icp_align_batch([gm] * count, scans[:count], guesses[:count], p)
        for a, c in zip(single, batch):
            assert a["n_iterations"] == c["n_iterations"] > 2 and a["termination_reason"] == c["termination_reason"]
            np.testing.assert_array_equal(a["T"], c["T"])
            np.testing.assert_array_equal(a["cov"], c["cov"])
            assert a["n_final_pairs"] == c["n_final_pairs"] and a["n_final_pairs_pt2pl"] == c["n_final_pairs_pt2pl"]


def test_lockstep_batch_of_row_kernel_layers(ctx, monkeypatch):
    """Layers of a few thousand points (what lidar3d-default.yaml feeds align()) in one mh_icp_align_batch: the row
    kernel with the fused first accumulation runs in lock step (one launch over all jobs); bitwise equal to single
    alignments and to the per-stream fallback, ragged sizes, jobs stalling at different iterations."""
    scene = synth.make_scene(4321, 80.0, 25)
    mp = synth.make_map(scene, 60000, 4321)
    gm = capi.Map(ctx, 1.0, 20).build(mp)
    pose = [1.5, -0.8, synth.SENSOR_H, 0.05, 0.004, -0.003]
    full = synth.make_scan(scene, pose, rings=32, azimuths=400, seed=99)
    thr, kp = synth.threshold_schedule(2.0, 300)
    p = capi.ICPParams(max_iterations=300, threshold=thr, kernel_param=kp, poll_every=5)
    rng = np.random.default_rng(8)
    sizes = [3000, 5000, 9000, 2500]
    ctxs = [capi.Context(0) for _ in sizes]
    scans, guesses, singles = [], [], []
    for c, n in zip(ctxs, sizes):
        sub = full[rng.permutation(len(full))[:n]]
        g = synth.pose_from_ypr(np.array(pose) + [0.3, 0.1, 0.02, 0.01, 0.002, 0.002] + rng.normal(0, 0.02, 6) * [1, 1, 0, 0, 0, 0])
        scans.append(capi.Scan(c, sub)); guesses.append(g)
        singles.append(capi.icp_align(gm, capi.Scan(ctx, sub), g, p, want_trace=False))
    batch = capi.icp_align_batch([gm] * len(sizes), scans, guesses, p)
    monkeypatch.setenv("MH_NO_LOCKSTEP", "1")
    streams = capi.icp_align_batch([gm] * len(sizes), scans, guesses, p)
    monkeypatch.delenv("MH_NO_LOCKSTEP")
    assert len({s["n_iterations"] for s in singles}) > 1
    for a, b, c in zip(singles, batch, streams):
        for r in (b, c):
            assert (r["n_iterations"], r["termination_reason"], r["n_final_pairs"]) == (a["n_iterations"], a["termination_reason"], a["n_final_pairs"])
            assert np.array_equal(r["T"], a["T"]) and np.array_equal(r["cov"], a["cov"])
    for c in ctxs:
        c.close()


def test_lockstep_small_chains_with_per_job_parameters(ctx, monkeypatch):
    """What N sequences on one GPU hand to one mh_icp_align_batch: layers of ~1-3 k points, every job with ITS OWN
    parameters (adaptive-threshold schedule, remaining iteration budget, hook check point, prior) and its own map.  Jobs
    with the same kernel chain advance in lock step -- the one-workgroup chain (<= 2 k points), the row kernel with the
    fused accumulation (above), and the one-workgroup chain with Matcher_Point2Plane on NDT maps -- several groups in one
    call; every result bitwise the single alignment's, as is the per-stream fallback's."""
    ws = [synth.make_workload("t", 60000, 32, 400, 80.0, 25, variant=v) for v in range(2)]
    maps = [capi.Map(ctx, 1.0, 20).build(w.map_xyz) for w in ws]
    ndt = [capi.Map(ctx, 1.0, 0, min_distance_between_points=0.05, ndt_max_eigen_ratio=0.05).build(w.map_xyz) for w in ws]
    rng = np.random.default_rng(11)
    sizes = [900, 1500, 2048, 3000, 1200, 5000, 700]
    jobs = []
    for k, n in enumerate(sizes):
        w = ws[k % 2]
        sub = w.scan_xyz[rng.permutation(len(w.scan_xyz))[:n]]
        sigma = [2.0, 1.2, 0.6, 2.0, 0.9, 1.5, 2.5][k]
        iters = [300, 120, 40, 300, 7, 300, 60][k]
        thr, kp = synth.threshold_schedule(sigma, iters)
        kw = dict(max_iterations=iters, threshold=thr, kernel_param=kp, poll_every=[0, 3, 5, 0, 2, 4, 0][k])
        if k == 1:
            kw.update(hook_enabled=True, hook_min_trans=0.15, hook_min_rot=float(np.deg2rad(0.75)), hook_checkpoint=w.T_guess)
        prior = (w.T_guess, np.diag([4.0, 4.0, 4.0, 100.0, 100.0, 100.0])) if k == 2 else None
        jobs.append((k % 2, sub, w.T_guess, capi.ICPParams(**kw), prior))
    ctxs = [capi.Context(0) for _ in jobs]
    scans = [capi.Scan(c, j[1]) for c, j in zip(ctxs, jobs)]
    for label, mset, pl in (("plain", maps, None), ("ndt", ndt, 0.4)):
        ps = []
        for j in jobs:
            q = j[3]
            if pl is not None:
                q = capi.ICPParams(**{**q.__dict__, "pt2pl_threshold": pl})
            ps.append(q)
        ms = [mset[j[0]] for j in jobs]
        singles = [capi.icp_align(m, capi.Scan(ctx, j[1]), j[2], q, prior=j[4], want_trace=False) for m, j, q in zip(ms, jobs, ps)]
        assert len({s["n_iterations"] for s in singles}) > 2 and any(capi.TERM_NAMES[s["termination_reason"]] == "HookRequest" for s in singles)
        batch = capi.icp_align_batch(ms, scans, [j[2] for j in jobs], ps, priors=[j[4] for j in jobs])
        monkeypatch.setenv("MH_NO_LOCKSTEP", "1")
        streams = capi.icp_align_batch(ms, scans, [j[2] for j in jobs], ps, priors=[j[4] for j in jobs])
        monkeypatch.delenv("MH_NO_LOCKSTEP")
        for a, b, c in zip(singles, batch, streams):
            for r in (b, c):
                assert (r["n_iterations"], r["termination_reason"], r["n_final_pairs"], r["n_final_pairs_pt2pl"]) == (
                    a["n_iterations"], a["termination_reason"], a["n_final_pairs"], a["n_final_pairs_pt2pl"]), label
                assert np.array_equal(r["T"], a["T"]) and np.array_equal(r["cov"], a["cov"]) and r["quality"] == a["quality"], label
        if pl is not None:
            assert all(s["n_final_pairs_pt2pl"] > 0 for s in singles)
    for c in ctxs:
        c.close()


def test_batch_pairs_block_pinned_uploads_and_distinct_maps(ctx, monkeypatch):
    """What bench.py's timed region does, at test size: every job has its OWN map (independent draws of the generator),
    the scans arrive through asynchronous uploads from page-locked host memory queued right before the batch (the batch
    has to order itself after them, whichever stream it runs on), and Results::finalPairings of every job come back in
    one pairs block -- pageable host, page-locked host (asynchronous download, complete after mh_ctx_synchronize of the
    first job's context) and device memory.  Lock-step chain (row kernel, ragged sizes) and the per-stream fallback:
    poses bitwise those of single alignments, pairings bit-equal to mh_icp_align's final_pairs."""
    import torch
    ws = [synth.make_workload("t", 60000, 32, 400, 80.0, 25, variant=v) for v in range(3)]
    maps = [capi.Map(ctx, 1.0, 20).build(w.map_xyz) for w in ws]
    sizes = [3000, 5000, 2600]
    thr, kp = synth.threshold_schedule(2.0, 40)
    p = capi.ICPParams(max_iterations=40, threshold=thr, kernel_param=kp, poll_every=5)
    rng = np.random.default_rng(3)
    subs = [w.scan_xyz[rng.permutation(len(w.scan_xyz))[:n]] for w, n in zip(ws, sizes)]
    guesses = [w.T_guess for w in ws]
    singles = [capi.icp_align(m, capi.Scan(ctx, sub), g, p, want_trace=False, want_pairs=True)
               for m, sub, g in zip(maps, subs, guesses)]
    assert all(s["n_final_pairs"] > 500 for s in singles)
    ctxs = [capi.Context(0) for _ in sizes]
    scans = [capi.Scan(c, np.zeros((n, 3), np.float32)) for c, n in zip(ctxs, sizes)]
    pinned = [torch.from_numpy(np.ascontiguousarray(sub.T)).pin_memory() for sub in subs]
    nbytes = sum(capi.pairs_block_bytes(n) for n in sizes)
    for env in (None, "MH_NO_LOCKSTEP"):
        if env:
            monkeypatch.setenv(env, "1")
        for mem in (capi.MEM_HOST, capi.MEM_HOST_PINNED, capi.MEM_DEVICE):
            for sc, t, n in zip(scans, pinned, sizes):  # asynchronous uploads, still in flight when the batch starts
                sc.update(np.zeros((n, 3), np.float32))
                sc.update_pinned(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), n)
            if mem == capi.MEM_DEVICE:
                dev = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
                torch.cuda.synchronize()
                res = capi.icp_align_batch(maps, scans, guesses, p, pairs_block=dev.data_ptr(), pairs_mem=mem)
                ctxs[0].synchronize()
                block = dev.cpu().numpy()
            else:
                host = torch.zeros(nbytes, dtype=torch.uint8)
                if mem == capi.MEM_HOST_PINNED:
                    host = host.pin_memory()
                res = capi.icp_align_batch(maps, scans, guesses, p, pairs_block=host.data_ptr(), pairs_mem=mem)
                ctxs[0].synchronize()  # (the pinned download completes here)
                block = host.numpy()
            for a, r, pr in zip(singles, res, capi.unpack_pairs_block(block, sizes, res)):
                assert (r["n_iterations"], r["termination_reason"], r["n_final_pairs"]) == (
                    a["n_iterations"], a["termination_reason"], a["n_final_pairs"])
                assert np.array_equal(r["T"], a["T"]) and np.array_equal(r["cov"], a["cov"])
                for k in ("local_idx", "global_idx", "global_xyz", "d2"):
                    assert np.array_equal(pr[k], a["pairs"][k]), (env, mem, k)
        if env:
            monkeypatch.delenv(env)
    for c in ctxs:
        c.close()


def test_batch_pairs_block_with_a_trivial_first_job(ctx):
    """job 0 is trivial (an empty scan, then max_iterations = 0 through per-job parameters), so the
    lock-step group's leader is ANOTHER job's context than the one that owns the pairs staging, the copy stream and its
    events: the page-locked download must still be queued (used to fail with MH_ERR_HIP after the alignments ran) and
    complete on mh_ctx_synchronize of the FIRST job's context; a second batch right behind it must wait for it."""
    import torch
    ws = [synth.make_workload("t", 60000, 32, 400, 80.0, 25, variant=v) for v in range(3)]
    maps = [capi.Map(ctx, 1.0, 20).build(w.map_xyz) for w in ws]
    thr, kp = synth.threshold_schedule(2.0, 30)
    p = capi.ICPParams(max_iterations=30, threshold=thr, kernel_param=kp, poll_every=5)
    rng = np.random.default_rng(5)
    subs = [np.zeros((0, 3), np.float32)] + [w.scan_xyz[rng.permutation(len(w.scan_xyz))[:n]] for w, n in zip(ws[1:], (3000, 4200))]
    sizes = [len(x) for x in subs]
    guesses = [w.T_guess for w in ws]
    singles = [None] + [capi.icp_align(m, capi.Scan(ctx, sub), g, p, want_trace=False, want_pairs=True)
                        for m, sub, g in zip(maps[1:], subs[1:], guesses[1:])]
    ctxs = [capi.Context(0) for _ in sizes]
    scans = [capi.Scan(c, sub) for c, sub in zip(ctxs, subs)]
    nbytes = sum(capi.pairs_block_bytes(n) for n in sizes)
    for mem in (capi.MEM_HOST_PINNED, capi.MEM_HOST):
        for rep in range(2):  # the second batch finds pairs_copy_pending set on the first job's context
            host = torch.zeros(nbytes, dtype=torch.uint8)
            if mem == capi.MEM_HOST_PINNED:
                host = host.pin_memory()
            res = capi.icp_align_batch(maps, scans, guesses, p, pairs_block=host.data_ptr(), pairs_mem=mem)
            ctxs[0].synchronize()
            assert res[0]["n_final_pairs"] == 0 and res[0]["termination_reason"] == 1  # NoPairings
            for a, r, pr in list(zip(singles, res, capi.unpack_pairs_block(host.numpy(), sizes, res)))[1:]:
                assert (r["n_iterations"], r["n_final_pairs"]) == (a["n_iterations"], a["n_final_pairs"]) and np.array_equal(r["T"], a["T"])
                for k in ("local_idx", "global_idx", "global_xyz", "d2"):
                    assert np.array_equal(pr[k], a["pairs"][k]), (mem, rep, k)
    # the other way to be trivial: a full-size job 0 with a budget of zero iterations (per-job parameters)
    scans[0].update(ws[0].scan_xyz[:2000])
    sizes[0] = 2000
    p0 = capi.ICPParams(max_iterations=0, threshold=thr[:1], kernel_param=kp[:1])
    nbytes = sum(capi.pairs_block_bytes(n) for n in sizes)
    host = torch.zeros(nbytes, dtype=torch.uint8).pin_memory()
    res = capi.icp_align_batch(maps, scans, guesses, [p0, p, p], pairs_block=host.data_ptr(), pairs_mem=capi.MEM_HOST_PINNED)
    ctxs[0].synchronize()
    assert res[0]["n_iterations"] == 0
    for a, r, pr in list(zip(singles, res, capi.unpack_pairs_block(host.numpy(), sizes, res)))[1:]:
        assert np.array_equal(r["T"], a["T"]) and np.array_equal(pr["global_idx"], a["pairs"]["global_idx"])
    for c in ctxs:
        c.close()


def test_align_is_bitwise_reproducible(ctx, small):
    w, gm, om, gs = small
    p = _params(capi, w, disable_stall_test=True)
    a = capi.icp_align(gm, gs, w.T_guess, p, want_trace=False)
    b = capi.icp_align(gm, gs, w.T_guess, p, want_trace=False)
    np.testing.assert_array_equal(a["T"], b["T"])
    np.testing.assert_array_equal(a["cov"], b["cov"])


@pytest.mark.parametrize("ndt", [False, True])
def test_small_layer_loop_controls_give_the_same_bits(ctx, ndt, monkeypatch):
    """k_step16's launches are driven by the state block alone, so how the host queues them must not matter: streaming control
    (a launch or two ahead of the published progress), chunks of launches closed by a one-workgroup launch and polled every
    1 / 3 / 7 iterations, the same chunks replayed as a hipGraph (captured when an alignment's shape repeats) -- one result."""
    pts = _ndt_cloud(51)
    gm = capi.Map(ctx, 1.0, 0, 0, 0.1, 0.05, 4).build(pts) if ndt else capi.Map(ctx, 1.0, 20).build(pts)
    rng = np.random.default_rng(52)
    n = 1700
    sub = pts[rng.integers(0, len(pts), n)] + rng.normal(0, 0.01, (n, 3)).astype(np.float32)
    guess = synth.pose_from_ypr([0.11, -0.07, 0.05, 0.006, -0.004, 0.01])
    thr, kp = synth.threshold_schedule(0.5, 60)
    kw = dict(max_iterations=60, threshold=thr, kernel_param=kp, gn=capi.GNParams(max_inner_iterations=2))
    if ndt:
        kw["pt2pl_threshold"] = 0.5
    scan = capi.Scan(ctx, sub)
    ref = capi.icp_align(gm, scan, guess, capi.ICPParams(**kw))  # streaming control
    assert ref["n_iterations"] > 4
    runs = []
    for poll in (1, 3, 7):
        runs.append(capi.icp_align(gm, scan, guess, capi.ICPParams(poll_every=poll, **kw)))
    monkeypatch.setenv("MH_NO_STREAM", "1")
    for _ in range(4):  # automatic chunks: the second call's chunk is captured, the later ones replay the graph
        runs.append(capi.icp_align(gm, scan, guess, capi.ICPParams(**kw)))
    monkeypatch.setenv("MH_NO_GRAPH", "1")
    runs.append(capi.icp_align(gm, scan, guess, capi.ICPParams(**kw)))
    for r in runs:
        assert r["n_iterations"] == ref["n_iterations"] and r["termination_reason"] == ref["termination_reason"]
        np.testing.assert_array_equal(r["T"], ref["T"])
        np.testing.assert_array_equal(r["cov"], ref["cov"])
        assert r["n_final_pairs"] == ref["n_final_pairs"]


@pytest.mark.parametrize("ndt", [False, True])
@pytest.mark.parametrize("search", ["rows", "plan_scan"])
def test_one_launch_loop_of_small_layers(ctx, ndt, search, monkeypatch):
    """Single alignments of layers up to 2560 points run their whole loop in ONE launch (k_icp16: the workgroups exchange the
    partial sums among themselves).  Same bits as the launch-by-launch chain (MH_NO_LOOP16=1); the loop is what runs by default
    and none is abandoned; when the device's admission limit is taken (MH_LOOP16_CUS) or a loop does not run to its end
    (MH_LOOP16_TEST_ABANDON) the chain gives the same result.  search = plan_scan (MH_LOOPW=all): the loop with the plan / scan
    search of mh_nn_flat.h, 128 points per workgroup, layers up to 4096 points (k_icpw; the default for lock-step batches)."""
    wave = search == "plan_scan"
    if wave and ndt:
        pytest.skip("NDT maps keep k_icp16<true>: the plane matcher rides in the row search")
    if wave:
        monkeypatch.setenv("MH_LOOPW", "all")
    pts = _ndt_cloud(61)
    gm = capi.Map(ctx, 1.0, 0, 0, 0.1, 0.05, 4).build(pts) if ndt else capi.Map(ctx, 1.0, 20).build(pts)
    rng = np.random.default_rng(62)
    thr, kp = synth.threshold_schedule(0.5, 60)
    kw = dict(max_iterations=60, threshold=thr, kernel_param=kp, gn=capi.GNParams(max_inner_iterations=2))
    if ndt:
        kw["pt2pl_threshold"] = 0.5
    p = capi.ICPParams(**kw)
    for n in (1, 31, 32, 33, 700, 1400, 2048, 2560) + ((2561, 3000, 4096) if wave else ()):
        sub = pts[rng.integers(0, len(pts), n)] + rng.normal(0, 0.01, (n, 3)).astype(np.float32)
        guess = synth.pose_from_ypr([0.11, -0.07, 0.05, 0.006, -0.004, 0.01])
        scan = capi.Scan(ctx, sub)
        s0, a0 = capi.loop_stats()
        loop = [capi.icp_align(gm, scan, guess, p, want_trace=True) for _ in range(3)]
        s1, a1 = capi.loop_stats()
        assert s1 - s0 == 3 and a1 == a0
        monkeypatch.setenv("MH_NO_LOOP16", "1")
        chain = capi.icp_align(gm, scan, guess, p, want_trace=True)
        assert capi.loop_stats() == (s1, a1)
        monkeypatch.delenv("MH_NO_LOOP16")
        monkeypatch.setenv("MH_LOOP16_CUS", "0")
        refused = capi.icp_align(gm, scan, guess, p, want_trace=True)
        assert capi.loop_stats() == (s1, a1)
        monkeypatch.delenv("MH_LOOP16_CUS")
        monkeypatch.setenv("MH_LOOP16_TEST_ABANDON", "1")
        again = capi.icp_align(gm, scan, guess, p, want_trace=True)
        assert capi.loop_stats() == (s1 + 1, a1 + 1)
        monkeypatch.delenv("MH_LOOP16_TEST_ABANDON")
        monkeypatch.setenv("MH_NO_LOOPW", "1")  # k_icp16 (a DPP row per point) where k_icpw (plan / scan search) is the default
        rows = capi.icp_align(gm, scan, guess, p, want_trace=True)
        assert capi.loop_stats() == (s1 + (2 if n <= 2560 else 1), a1 + 1)  # (k_icp16 takes layers up to 2560 points)
        if wave:
            monkeypatch.setenv("MH_LOOPW", "all")
        monkeypatch.delenv("MH_NO_LOOPW")
        for r in loop + [refused, again, rows]:
            assert r["n_iterations"] == chain["n_iterations"] and r["termination_reason"] == chain["termination_reason"]
            np.testing.assert_array_equal(r["T"], chain["T"])
            np.testing.assert_array_equal(r["cov"], chain["cov"])
            assert r["n_final_pairs"] == chain["n_final_pairs"] and r["n_final_pairs_pt2pl"] == chain["n_final_pairs_pt2pl"]
            for a, b in zip(r["trace"], chain["trace"]):
                np.testing.assert_array_equal(a["T"], b["T"])


@pytest.mark.parametrize("ndt", [False, True])
@pytest.mark.parametrize("search", ["default", "rows"])
def test_one_launch_loops_of_a_lockstep_batch(ctx, ndt, search, monkeypatch):
    """A lock-step batch of small layers runs the jobs' whole loops side by side in ONE launch -- point layers: k_icpw_b, the plan /
    scan search, every job its own workgroups of 128 points; NDT maps and search = rows (MH_LOOPW=none): k_icp16_b, a job's
    workgroups take several groups of 32 points when the jobs have to share the CUs.  Same bits as single alignments, as the
    launch-by-launch batch (MH_NO_LOOP16_BATCH=1) and as the second attempt after an abandoned loop."""
    if search == "rows":
        if ndt:
            pytest.skip("NDT maps run k_icp16_b either way")
        monkeypatch.setenv("MH_LOOPW", "none")
    pts = _ndt_cloud(81)
    gm = capi.Map(ctx, 1.0, 0, 0, 0.1, 0.05, 4).build(pts) if ndt else capi.Map(ctx, 1.0, 20).build(pts)
    rng = np.random.default_rng(82)
    thr, kp = synth.threshold_schedule(0.5, 60)
    kw = dict(max_iterations=60, threshold=thr, kernel_param=kp, gn=capi.GNParams(max_inner_iterations=2))
    if ndt:
        kw["pt2pl_threshold"] = 0.5
    p = capi.ICPParams(**kw)
    sizes = [1500, 900, 2048, 33, 1, 1400, 1400, 700, 2000, 1999, 64, 1234]  # 12 jobs: 21 workgroups each, up to 4 groups per workgroup
    ctxs = [capi.Context(0) for _ in sizes]
    subs = [pts[rng.integers(0, len(pts), n)] + rng.normal(0, 0.01, (n, 3)).astype(np.float32) for n in sizes]
    guesses = [synth.pose_from_ypr([0.1 + 0.02 * (k % 4), -0.08, 0.05, 0.006, -0.004, 0.01]) for k in range(len(sizes))]
    single = [capi.icp_align(gm, capi.Scan(ctx, s), g, p, want_trace=False) for s, g in zip(subs, guesses)]
    scans = [capi.Scan(c, s) for c, s in zip(ctxs, subs)]

    def same(batch, count):
        for a, c in zip(single[:count], batch):
            assert a["n_iterations"] == c["n_iterations"] and a["termination_reason"] == c["termination_reason"]
            np.testing.assert_array_equal(a["T"], c["T"])
            np.testing.assert_array_equal(a["cov"], c["cov"])
            assert a["n_final_pairs"] == c["n_final_pairs"] and a["n_final_pairs_pt2pl"] == c["n_final_pairs_pt2pl"]

    for count in (len(sizes), 8, 2):
        s0, a0 = capi.loop_stats()
        same(capi.icp_align_batch([gm] * count, scans[:count], guesses[:count], p), count)
        s1, a1 = capi.loop_stats()
        assert s1 - s0 == count and a1 == a0  # every job ran as a loop
        monkeypatch.setenv("MH_NO_LOOP16_BATCH", "1")
        same(capi.icp_align_batch([gm] * count, scans[:count], guesses[:count], p), count)
        assert capi.loop_stats() == (s1, a1)
        monkeypatch.delenv("MH_NO_LOOP16_BATCH")
        monkeypatch.setenv("MH_LOOP16_TEST_ABANDON", "1")
        same(capi.icp_align_batch([gm] * count, scans[:count], guesses[:count], p), count)
        s2, a2 = capi.loop_stats()
        assert s2 - s1 == count and a2 - a1 >= 1
        monkeypatch.delenv("MH_LOOP16_TEST_ABANDON")
    for c in ctxs:
        c.close()


def test_solo_hint_agrees_with_what_a_single_alignment_does(ctx, monkeypatch):
    """mh_icp_align_prefers_solo (the batching hint of the multi-sequence runner) says "one launch" exactly when mh_icp_align
    then starts a one-launch loop."""
    pts = _ndt_cloud(71)
    gm = capi.Map(ctx, 1.0, 20).build(pts)
    rng = np.random.default_rng(72)
    thr, kp = synth.threshold_schedule(0.5, 12)
    guess = synth.pose_from_ypr([0.05, -0.03, 0.02, 0.003, -0.002, 0.005])
    cases = [(n, dict(), dict()) for n in (1, 500, 2560, 2561, 4096, 4097, 6000)]
    cases += [(500, dict(poll_every=3), dict()), (500, dict(profile=1), dict()), (500, dict(), {"MH_NO_LOOP16": "1"}),
              (500, dict(), {"MH_NO_STREAM": "1"}), (500, dict(), {"MH_MATCH": "q"}), (40000, dict(), {"MH_MATCH": "s"})]
    cases += [(n, dict(), {"MH_LOOPW": "all"}) for n in (500, 2561, 4096, 4097)]  # k_icpw also for single alignments
    for n, extra, env in cases:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        p = capi.ICPParams(max_iterations=12, threshold=thr, kernel_param=kp, **extra)
        sub = pts[rng.integers(0, len(pts), n)] + rng.normal(0, 0.01, (n, 3)).astype(np.float32)
        scan = capi.Scan(ctx, sub)
        hint = capi.icp_align_prefers_solo(scan, p, guess)
        s0, a0 = capi.loop_stats()
        capi.icp_align(gm, scan, guess, p, want_trace=False)
        s1, a1 = capi.loop_stats()
        assert (s1 - s0 == 1) == hint and a1 == a0, (n, extra, env, hint, s1 - s0)
        wave = env.get("MH_LOOPW") == "all"
        assert hint == (n <= (4096 if wave else 2560) and not extra and (not env or wave))
        if hint:  # ... with company: as long as everybody's workgroups fit the device together -- and never more than four callers
            groups = (n + 31) // 32
            units = (groups + 3) // 4 if wave else 2 * groups  # half CUs: a k_icpw workgroup of 128 points one, a k_icp16 workgroup of 32 points two
            def fits(callers):  # (MI355X: 256 CUs = 512 half CUs, 70 % of them for loops)
                return callers <= 1 or (callers <= 4 and callers * units * 10 <= 512 * 7)
            for callers in (1, 2, 3, 4, 5, 8, 9, 16, 17, 64, 1000):
                assert capi.icp_align_prefers_solo(scan, p, guess, concurrent_callers=callers) == fits(callers), (n, callers)
        for k in env:
            monkeypatch.delenv(k)


def test_scan_update_reuses_handle(ctx, oracle, small):
    w, gm, om, gs = small
    s = capi.Scan(ctx, w.scan_xyz[:100])
    s.update(w.scan_xyz[:1777])
    assert len(s) == 1777
    g = capi.nn_search(gm, s, w.T_guess, 1.0)
    o = oracle.match_points(om, w.scan_xyz[:1777], w.T_guess, 1.0)
    np.testing.assert_array_equal(g["global_idx"], o["global_idx"])


def test_profile_fields(ctx, small):
    w, gm, om, gs = small
    r = capi.icp_align(gm, gs, w.T_guess, _params(capi, w, disable_stall_test=True, profile=True), want_trace=False)
    assert r["n_match_launches"] == w.n_iters and r["match_kernel_ms"] > 0 and r["total_ms"] >= r["match_kernel_ms"]


@pytest.mark.parametrize("env", [{"MH_MATCH": "s"}, {"MH_MATCH": "s", "MH_NO_STEP_CHAIN": "1"},
                                 {"MH_MATCH": "s", "MH_NO_STEP_CHAIN": "1", "MH_NO_FUSE16": "1"}, {"MH_MATCH": "q"},
                                 {"MH_MATCH": "p"}, {"MH_MATCH": "x"}, {"MH_MATCH": "q", "MH_NO_GRAPH": "1"},
                                 {"MH_MATCH": "t"}, {"MH_MATCH": "t", "MH_NO_GRAPH": "1"}, {"MH_MATCH": "w"},
                                 {"MH_MATCH": "w", "MH_NO_GRAPH": "1"}, {"MH_MATCH": "w", "MH_WAVE_LDS": "1"},
                                 {"MH_MATCH": "o"}, {"MH_MATCH": "f"}, {"MH_MATCH": "f", "MH_NO_GRAPH": "1"}])
@pytest.mark.parametrize("n_scan", [2000, 5000])
def test_every_kernel_variant_matches_the_oracle(ctx, oracle, env, n_scan, monkeypatch):
    """The default path picks its kernels by layer size (row / quad search, one-workgroup or multi-launch solve);
    here every combination is forced on the same inputs: identical pairings and termination, poses within 1e-9."""
    scene = synth.make_scene(4321, 80.0, 25)
    mp = synth.make_map(scene, 60000, 4321)
    pose = [1.5, -0.8, synth.SENSOR_H, 0.05, 0.004, -0.003]
    scan = synth.make_scan(scene, pose, rings=32, azimuths=400, seed=99)[:n_scan]
    guess = synth.pose_from_ypr(np.array(pose) + [0.3, 0.1, 0.02, 0.01, 0.002, 0.002])
    thr, kp = synth.threshold_schedule(2.0, 300)
    kw = dict(max_iterations=300, threshold=thr, kernel_param=kp)
    o = oracle.icp_align(oracle.Map(1.0, 20).insert(mp), scan, guess, oracle.ICPParams(**kw), want_pairs=True)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    g = capi.icp_align(capi.Map(ctx, 1.0, 20).build(mp), capi.Scan(ctx, scan), guess, capi.ICPParams(**kw), want_pairs=True)
    assert g["n_iterations"] == o["n_iterations"] and g["termination_reason"] == o["termination_reason"]
    assert [t["n_pairs"] for t in g["trace"]] == [t["n_pairs"] for t in o["trace"]]
    np.testing.assert_array_equal(g["pairs"]["global_idx"], o["pairs"]["global_idx"])
    np.testing.assert_array_equal(g["pairs"]["d2"], o["pairs"]["d2"])
    np.testing.assert_allclose(g["T"], o["T"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("vs,shift", [(0.25, 0.9), (0.4, 0.6), (1.0, 0.3)])
def test_previous_pairing_bound_and_its_fallback(ctx, oracle, vs, shift, monkeypatch):
    """From the second iteration on the quad matcher bounds its search by the distance to the record a point was paired with
    before (nn_search_quad, bound0).  Small voxels and a guess that is several voxels off: between iterations points
    move by more than a voxel, the old partner leaves the 27-voxel block, the bound is not attained inside it and the
    search has to run again without one -- per-iteration pair counts, final pairings and d2 stay the oracle's, bit for bit,
    with the bound and without (MH_NO_PREV_BOUND)."""
    scene = synth.make_scene(777, 60.0, 20)
    mp = synth.make_map(scene, 150000, 777)
    pose = [0.7, -0.4, synth.SENSOR_H, 0.03, 0.002, -0.002]
    scan = synth.make_scan(scene, pose, rings=48, azimuths=500, seed=5)[:20000]
    guess = synth.pose_from_ypr(np.array(pose) + [shift, -0.5 * shift, 0.05, 0.02, 0.003, 0.002])
    thr, kp = synth.threshold_schedule(2.0, 40)
    kw = dict(max_iterations=40, threshold=thr, kernel_param=kp)
    o = oracle.icp_align(oracle.Map(vs, 20).insert(mp), scan, guess, oracle.ICPParams(**kw), want_pairs=True)
    gm, gs = capi.Map(ctx, vs, 20).build(mp), capi.Scan(ctx, scan)
    for match, no_bound in (("q", False), ("f", False), ("q", True), ("f", True)):  # f: the plan / scan matcher (phases C -> D: "not attained")
        monkeypatch.setenv("MH_MATCH", match)
        if no_bound:
            monkeypatch.setenv("MH_NO_PREV_BOUND", "1")
        g = capi.icp_align(gm, gs, guess, capi.ICPParams(**kw), want_pairs=True)
        assert g["n_iterations"] == o["n_iterations"] and g["termination_reason"] == o["termination_reason"]
        assert [t["n_pairs"] for t in g["trace"]] == [t["n_pairs"] for t in o["trace"]]
        for k in ("local_idx", "global_idx", "d2", "global_xyz"):
            np.testing.assert_array_equal(g["pairs"][k], o["pairs"][k])
        np.testing.assert_allclose(g["T"], o["T"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("vs,cap,mode,lattice", [(1.0, 20, 0, False), (0.5, 31, 0, True), (1.0, 48, 0, False), (1.0, 20, 1, False),
                                                 (0.7, 8, 0, True)])
def test_quad_matcher_sub_voxel_index(ctx, oracle, vs, cap, mode, lattice, monkeypatch):
    """The quad matcher scans a voxel's records in the order of its sub-voxel index (MapView::pts_q: stably re-ordered by
    (x half, y half) of the voxel, the scan position in w) and, under a bound, only the hull of the quadrants that can hold a
    record within it (qidx: the quadrants' boundaries at the voxel's hash slot, built lazily after every rebuild).  Voxels with
    more than 31 records and trunc-indexed maps have no boundaries; MH_NO_QIDX=1 gives none to anybody.  Per-iteration pair
    counts, final pairings and d2 are the oracle's bit for bit -- on a lattice map (exact ties inside and across quadrants,
    records ON the mid planes), with the index and without, before and after a key-frame insertion rebuilds the records."""
    rng = np.random.default_rng(int(vs * 10) + cap + mode)
    scene = synth.make_scene(4711, 60.0, 20)
    mp = synth.make_map(scene, 120000, 4711)
    if lattice:
        mp = (np.round(mp / np.float32(vs / 4)) * np.float32(vs / 4)).astype(np.float32)  # quarter-voxel lattice: mid planes hit
    pose = [0.7, -0.4, synth.SENSOR_H, 0.03, 0.002, -0.002]
    scan = synth.make_scan(scene, pose, rings=64, azimuths=700, seed=5)
    scan = scan[rng.permutation(len(scan))[:36000]]   # above the row matcher's layer size: the plan / scan matcher by default (round 5), whose phase D is the quad matcher
    guess = synth.pose_from_ypr(np.array(pose) + [0.25, -0.15, 0.02, 0.01, 0.002, 0.001])
    thr, kp = synth.threshold_schedule(2.0, 12)
    kw = dict(max_iterations=12, threshold=thr, kernel_param=kp, disable_stall_test=True)
    om = oracle.Map(vs, cap, mode).insert(mp)
    extra = synth.make_scan(scene, [3.0, 1.0, synth.SENSOR_H, 0.2, 0.0, 0.0], rings=32, azimuths=400, seed=8)
    I = np.eye(4)[:3]
    gs = capi.Scan(ctx, scan)
    refs = [oracle.icp_align(om, scan, guess, oracle.ICPParams(**kw), want_pairs=True)]
    om.insert_posed(extra, I, 1000.0)
    refs.append(oracle.icp_align(om, scan, guess, oracle.ICPParams(**kw), want_pairs=True))
    for no_index in (False, True):
        if no_index:
            monkeypatch.setenv("MH_NO_QIDX", "1")
        gm = capi.Map(ctx, vs, cap, mode).build(mp)
        for step, o in enumerate(refs):
            if step == 1:
                gm.insert(capi.Scan(ctx, extra), I, 1000.0)
            g = capi.icp_align(gm, gs, guess, capi.ICPParams(**kw), want_pairs=True)
            assert [t["n_pairs"] for t in g["trace"]] == [t["n_pairs"] for t in o["trace"]]
            for k in ("local_idx", "global_idx", "d2", "global_xyz"):
                np.testing.assert_array_equal(g["pairs"][k], o["pairs"][k])
            np.testing.assert_allclose(g["T"], o["T"], rtol=0, atol=1e-9)
        # the index has re-written the count words of the map's hash slots (boundaries beside the count): every OTHER reader of
        # the table -- the one-lane matcher, the k-best matcher, the row matcher of a small layer's alignment -- still sees
        # the counts (slot_count()); `om` holds the key-frame by now, like the device map
        small = scan[:3000]
        a, b = capi.nn_search(gm, capi.Scan(ctx, small), guess, 1.5 * vs), oracle.match_points(om, small, guess, 1.5 * vs)
        for k in ("local_idx", "global_idx", "d2"):
            np.testing.assert_array_equal(a[k], b[k])
        a, b = capi.nn_search_k(gm, capi.Scan(ctx, small), guess, 1.5 * vs, 3), oracle.match_points_k(om, small, guess, 1.5 * vs, 3)
        for k in ("local_idx", "global_idx", "d2"):
            np.testing.assert_array_equal(a[k], b[k])
        g = capi.icp_align(gm, capi.Scan(ctx, small), guess, capi.ICPParams(**kw), want_pairs=True)
        o = oracle.icp_align(om, small, guess, oracle.ICPParams(**kw), want_pairs=True)
        assert [t["n_pairs"] for t in g["trace"]] == [t["n_pairs"] for t in o["trace"]]
        np.testing.assert_array_equal(g["pairs"]["global_idx"], o["pairs"]["global_idx"])


def test_sub_voxel_index_built_once_for_concurrent_searchers(oracle):
    """map_ensure_qidx builds the quad matcher's index on first use, on the stream of whoever asks first; alignments on OTHER
    contexts' streams -- here four host threads, a context and a scan each, one shared map, all starting at once on a map
    whose index does not exist yet -- order themselves behind that build.  Every thread's result is the serial one, bit for
    bit, and the oracle's pairing counts."""
    import threading
    scene = synth.make_scene(99, 60.0, 20)
    mp = synth.make_map(scene, 150000, 99)
    poses = [[0.5 * k, -0.3 * k, synth.SENSOR_H, 0.02 * k, 0.001, -0.002] for k in range(4)]
    scans = [synth.make_scan(scene, p, rings=64, azimuths=640, seed=40 + k)[:35000] for k, p in enumerate(poses)]
    guesses = [synth.pose_from_ypr(np.array(p) + [0.2, -0.1, 0.01, 0.008, 0.001, 0.001]) for p in poses]
    thr, kp = synth.threshold_schedule(2.0, 8)
    kw = dict(max_iterations=8, threshold=thr, kernel_param=kp, disable_stall_test=True)
    map_ctx = capi.Context(0)
    ctxs = [capi.Context(0) for _ in scans]
    gs = [capi.Scan(c, s) for c, s in zip(ctxs, scans)]
    for rnd in range(3):
        gm = capi.Map(map_ctx, 1.0, 20).build(mp)       # a fresh map: no index yet
        out = [None] * len(scans)

        def work(k):
            out[k] = capi.icp_align(gm, gs[k], guesses[k], capi.ICPParams(**kw), want_pairs=True)

        ts = [threading.Thread(target=work, args=(k,)) for k in range(len(scans))]
        [t.start() for t in ts]
        [t.join() for t in ts]
        for k in range(len(scans)):
            ref = capi.icp_align(gm, gs[k], guesses[k], capi.ICPParams(**kw), want_pairs=True)  # serial, index in place
            assert out[k]["T"].tobytes() == ref["T"].tobytes()
            for key in ("local_idx", "global_idx", "d2"):
                np.testing.assert_array_equal(out[k]["pairs"][key], ref["pairs"][key])
            if rnd == 0:
                o = oracle.icp_align(oracle.Map(1.0, 20).insert(mp), scans[k], guesses[k], oracle.ICPParams(**kw), want_pairs=True)
                assert [t["n_pairs"] for t in ref["trace"]] == [t["n_pairs"] for t in o["trace"]]
                np.testing.assert_array_equal(ref["pairs"]["global_idx"], o["pairs"]["global_idx"])


@pytest.mark.parametrize("n_scan,env", [(900, {}), (3000, {}), (3000, {"MH_NO_FUSE16": "1"}), (3000, {"MH_MATCH": "p"}),
                                        (9000, {}), (20000, {})])
def test_converged_alignment_with_early_inner_exit(ctx, oracle, n_scan, env, monkeypatch):
    """Stall test off and far more iterations than the alignment needs: once the Gauss-Newton step falls below min_delta
    the solver leaves its inner loop after the FIRST step of every ICP iteration.  The launches of the skipped inner step
    must then do nothing -- k_accum did, the k_solve behind it kept solving on stale partial sums (the fused matchers'
    wider layout read as k_accum's: a converged alignment jumped by metres; found by tools/fuzz_batch.py).  Every kernel
    chain (one workgroup, fused row kernel, row + k_accum, one lane per point, quad) against the oracle, iteration by
    iteration."""
    scene = synth.make_scene(4242, 70.0, 20)
    mp = synth.make_map(scene, 120000, 4242)
    pose = [1.0, -0.5, synth.SENSOR_H, 0.04, 0.003, -0.002]
    scan = synth.make_scan(scene, pose, rings=64, azimuths=1000, seed=9)
    scan = scan[np.random.default_rng(3).permutation(len(scan))[:n_scan]]
    guess = synth.pose_from_ypr(np.array(pose) + [-0.07, 0.28, 0.01, 0.004, -0.003, 0.002])
    thr, kp = synth.threshold_schedule(2.0, 60)
    kw = dict(max_iterations=60, threshold=thr, kernel_param=kp, disable_stall_test=True)
    o = oracle.icp_align(oracle.Map(1.0, 20).insert(mp), scan, guess, oracle.ICPParams(**kw), want_pairs=True)
    assert min(t["delta_trans"] for t in o["trace"]) < 1e-7  # (the regime this test is about)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    g = capi.icp_align(capi.Map(ctx, 1.0, 20).build(mp), capi.Scan(ctx, scan), guess, capi.ICPParams(**kw), want_pairs=True)
    assert g["n_iterations"] == o["n_iterations"] == 60 and g["termination_reason"] == o["termination_reason"]
    assert [t["n_pairs"] for t in g["trace"]] == [t["n_pairs"] for t in o["trace"]]
    for a, b in zip(g["trace"], o["trace"]):
        np.testing.assert_allclose(a["T"], b["T"], rtol=0, atol=1e-9)
    for k in ("local_idx", "global_idx", "d2", "global_xyz"):
        np.testing.assert_array_equal(g["pairs"][k], o["pairs"][k])
    np.testing.assert_allclose(g["T"], o["T"], rtol=0, atol=1e-9)
