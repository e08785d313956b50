"""-m gpu: device-side scan pre-processing and the device-resident incremental local map (row f2),
through the C ABI, against the CPU oracle.  Index sets, voxel contents and source indices are bit-exact; de-skewed
coordinates are floating point (fp64 sin/cos differ by an ulp between libm and the device) and are held to 1 float ulp."""
import numpy as np
import pytest

from mola_lidar_odometry_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _raw(seed, small_workload, with_nan=True):
    rng = np.random.default_rng(seed)
    xyz = small_workload.scan_xyz.copy()
    if with_nan:
        xyz[11] = [np.nan, 1, 1]
        xyz[500] = [1, -np.inf, 1]
    t = (1.7e4 + np.sort(rng.uniform(0.0, 0.1, len(xyz)))).astype(np.float32)
    return xyz, t


PP = dict(decim_map_resolution=0.35, decim_icp_resolution=1.1, min_points_to_filter=300, range_min=2.0, range_max=70.0,
          bbox_mode=1, bbox_min=(-8.0, -8.0, -1.8), bbox_max=(8.0, 8.0, 4.0))


@pytest.mark.parametrize("mode", [0, 1])
def test_preprocess_bit_exact(ctx, oracle, small_workload, mode):
    xyz, t = _raw(1, small_workload)
    raw = capi.Scan(ctx, xyz).set_timestamps(t)
    om, oi = capi.Scan(ctx), capi.Scan(ctx)
    raw.preprocess(capi.preprocess_params(index_mode=mode, timestamp_method=capi.TS_MIDDLE_IS_ZERO, time_offset=0.01, **PP),
                   om, oi)
    im, ii = oracle.preprocess(xyz, index_mode=mode, **PP)
    ta = oracle.adjust_timestamps(t, oracle.TS_MIDDLE_IS_ZERO, 0.01)
    assert 0 < len(ii) < len(im) < len(xyz)
    for scan, idx in ((om, im), (oi, ii)):
        d = scan.download()
        assert scan.n == len(idx)
        np.testing.assert_array_equal(d["src_idx"], idx)
        np.testing.assert_array_equal(d["xyz"], xyz[idx])
        np.testing.assert_array_equal(d["t"], ta[idx])


@pytest.mark.parametrize("methods", [(1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("mode", [0, 1])
def test_preprocess_closest_to_average_bit_exact(ctx, oracle, small_workload, methods, mode):
    """decimate_method: DecimateMethod::ClosestToAverage in either decimation of the chain (the commented alternative of
    lidar3d-default.yaml:292, rgbd.yaml:254-278): per voxel the point closest to the float mean accumulated in input order -- the
    device sorts a scan by voxel key (stable) and walks every voxel's run; same index sets as the oracle, also with exact ties
    (points on a lattice), non-finite points, the second stage below minimum_input_points_to_filter, and in a batch."""
    xyz, t = _raw(7, small_workload)
    xyz[::5] = np.round(xyz[::5] * 4) / 4
    mm, mi = methods
    for pp in (PP, dict(PP, decim_map_resolution=1.0, min_points_to_filter=1500)):  # (the second: stage 2 passes its input through)
        raw = capi.Scan(ctx, xyz).set_timestamps(t)
        om, oi = capi.Scan(ctx), capi.Scan(ctx)
        raw.preprocess(capi.preprocess_params(index_mode=mode, timestamp_method=capi.TS_MIDDLE_IS_ZERO, time_offset=0.01,
                                              decim_map_method=mm, decim_icp_method=mi, **pp), om, oi)
        im, ii = oracle.preprocess(xyz, index_mode=mode, decim_map_method=mm, decim_icp_method=mi, **pp)
        fm, fi = oracle.preprocess(xyz, index_mode=mode, **pp)
        assert 0 < len(ii) <= len(im) < len(xyz)
        if pp is PP:
            assert not np.array_equal(im, fm) or not np.array_equal(ii, fi)  # (not a no-op)
        else:
            assert len(ii) == len(im) < 1500  # (stage 2 left its input alone)
        for scan, idx in ((om, im), (oi, ii)):
            d = scan.download()
            assert scan.n == len(idx)
            np.testing.assert_array_equal(d["src_idx"], idx)
            np.testing.assert_array_equal(d["xyz"], xyz[idx])
    # a batch of ragged scans with the method per scan = single calls
    rng = np.random.default_rng(8)
    sizes = [len(xyz), len(xyz) // 3, 250, 0, 1500]
    ctxs = [capi.Context(0) for _ in sizes]
    raws, pars = [], []
    for k, n in enumerate(sizes):
        sel = np.sort(rng.choice(len(xyz), n, replace=False))
        raws.append(capi.Scan(ctxs[k], xyz[sel]))
        pars.append(capi.preprocess_params(index_mode=mode, decim_map_method=(mm, 0, 1)[k % 3], decim_icp_method=(mi, 1, 0)[k % 3], **PP))
    oms, ois = [capi.Scan(c) for c in ctxs], [capi.Scan(c) for c in ctxs]
    capi.preprocess_batch(raws, pars, oms, ois)
    for k in range(len(sizes)):
        sm, si = capi.Scan(ctxs[k]), capi.Scan(ctxs[k])
        raws[k].preprocess(pars[k], sm, si)
        for got, want in ((oms[k], sm), (ois[k], si)):
            np.testing.assert_array_equal(got.download()["src_idx"], want.download()["src_idx"])
    with pytest.raises(capi.MolahipError):
        raws[0].preprocess(capi.preprocess_params(decim_map_method=2, **PP), oms[0], ois[0])
    for c in ctxs:
        c.close()


def test_preprocess_edge_cases(ctx, oracle, small_workload):
    xyz, t = _raw(2, small_workload, with_nan=False)
    # smaller than minimum_input_points_to_filter: no decimation, predicates still apply; no time stamps attached
    few = xyz[:250]
    raw = capi.Scan(ctx, few)
    om, oi = capi.Scan(ctx), capi.Scan(ctx)
    raw.preprocess(capi.preprocess_params(**PP), om, oi)
    im, ii = oracle.preprocess(few, **PP)
    np.testing.assert_array_equal(om.download()["src_idx"], im)
    np.testing.assert_array_equal(oi.download()["src_idx"], ii)
    np.testing.assert_array_equal(om.download()["t"], np.zeros(len(im), np.float32))
    # all stages off: pass-through; EarliestIsZero
    raw = capi.Scan(ctx, xyz).set_timestamps(t)
    raw.preprocess(capi.preprocess_params(0.0, 0.0, timestamp_method=capi.TS_EARLIEST_IS_ZERO), om, None)
    d = om.download()
    np.testing.assert_array_equal(d["xyz"], xyz)
    np.testing.assert_array_equal(d["t"], oracle.adjust_timestamps(t, oracle.TS_EARLIEST_IS_ZERO))
    # empty input
    raw = capi.Scan(ctx)
    raw.preprocess(capi.preprocess_params(**PP), om, oi)
    assert om.n == 0 and oi.n == 0
    # everything filtered out
    raw = capi.Scan(ctx, xyz)
    raw.preprocess(capi.preprocess_params(0.3, 1.0, range_min=1e4, range_max=2e4), om, oi)
    assert om.n == 0 and oi.n == 0
    with pytest.raises(capi.MolahipError):
        raw.preprocess(capi.preprocess_params(**PP), om, om)
    with pytest.raises(capi.MolahipError):
        capi.Scan(ctx, xyz).set_timestamps(t[:10])


def test_preprocess_batch_equals_single_calls(ctx, oracle, small_workload):
    """mh_scan_preprocess_batch over ragged scans in their own contexts, per-scan parameters, one scan below
    minimum_input_points_to_filter, one empty, one without time stamps, one without an ICP layer: every output bit for bit
    what a single call gives (and the oracle)."""
    rng = np.random.default_rng(5)
    xyz, t = _raw(3, small_workload)
    sizes = [len(xyz), len(xyz) // 2, 250, 0, len(xyz) // 3, min(777, len(xyz))]
    ctxs = [capi.Context(0) for _ in sizes]
    pars, raws, clouds = [], [], []
    for k, n in enumerate(sizes):
        sel = np.sort(rng.choice(len(xyz), n, replace=False))
        c = xyz[sel] + np.float32(0.01 * k)
        raw = capi.Scan(ctxs[k], c)
        if k != 4 and n:
            raw.set_timestamps(t[sel])
        raws.append(raw)
        clouds.append(c)
        pars.append(capi.preprocess_params(decim_map_resolution=0.25 + 0.05 * k, decim_icp_resolution=0.9 + 0.1 * k, min_points_to_filter=300,
                                           range_min=1.0 + 0.3 * k, range_max=60.0 + k, bbox_mode=1 + (k % 2), bbox_min=(-8.0, -8.0, -1.8),
                                           bbox_max=(8.0 + k, 8.0, 4.0), index_mode=k % 2,
                                           timestamp_method=(capi.TS_MIDDLE_IS_ZERO, capi.TS_EARLIEST_IS_ZERO, capi.TS_NONE)[k % 3], time_offset=0.01 * k))
    oms = [capi.Scan(c) for c in ctxs]
    ois = [capi.Scan(c) if k != 5 else None for k, c in enumerate(ctxs)]
    capi.preprocess_batch(raws, pars, oms, ois)
    for k in range(len(sizes)):
        sm, si = capi.Scan(ctxs[k]), capi.Scan(ctxs[k])
        raws[k].preprocess(pars[k], sm, si if ois[k] is not None else None)
        for got, want in ((oms[k], sm), (ois[k], si)):
            if got is None:
                continue
            assert got.n == want.n
            a, b = got.download(), want.download()
            for key in ("xyz", "t", "src_idx"):
                np.testing.assert_array_equal(a[key], b[key])
        p = pars[k]
        im, ii = oracle.preprocess(clouds[k], decim_map_resolution=p.decim_map_resolution, decim_icp_resolution=p.decim_icp_resolution,
                                   min_points_to_filter=300, range_min=p.range_min, range_max=p.range_max, bbox_mode=p.bbox_mode,
                                   bbox_min=tuple(p.bbox_min), bbox_max=tuple(p.bbox_max), index_mode=p.index_mode)
        np.testing.assert_array_equal(oms[k].download()["src_idx"], im)
        if ois[k] is not None:
            np.testing.assert_array_equal(ois[k].download()["src_idx"], ii)
    # one parameter set for all; no ICP layers at all; the same call again (buffers reused)
    for _ in range(2):
        capi.preprocess_batch(raws, pars[1], oms, None)
        for k in range(len(sizes)):
            sm = capi.Scan(ctxs[k])
            raws[k].preprocess(pars[1], sm, None)
            np.testing.assert_array_equal(oms[k].download()["src_idx"], sm.download()["src_idx"])
    # an output listed twice is refused
    with pytest.raises(capi.MolahipError):
        capi.preprocess_batch(raws[:2], pars[0], [oms[0], oms[0]], None)
    for c in ctxs:
        c.close()




def test_deskew(ctx, oracle, small_workload):
    xyz, t = _raw(3, small_workload, with_nan=False)
    raw = capi.Scan(ctx, xyz).set_timestamps(t)
    sk, out = capi.Scan(ctx), capi.Scan(ctx)
    raw.preprocess(capi.preprocess_params(0.35, 0.0, timestamp_method=capi.TS_MIDDLE_IS_ZERO), sk, None)
    d0 = sk.download()
    tw = np.array([14.0, -0.3, 0.05, 0.01, -0.02, 0.7])
    ref = oracle.deskew(d0["xyz"], d0["t"], tw)
    sk.deskew(tw, out)
    d = out.download()
    np.testing.assert_array_equal(d["src_idx"], d0["src_idx"])
    np.testing.assert_array_equal(d["t"], d0["t"])
    ulp = np.spacing(np.abs(ref).astype(np.float32))
    assert np.all(np.abs(d["xyz"] - ref) <= ulp)
    assert np.mean(d["xyz"] == ref) > 0.999
    assert np.abs(d["xyz"] - d0["xyz"]).max() > 0.5  # the twist really moved points
    # the skewed layer is untouched and can be de-skewed again with another twist (LidarOdometry.cpp:992-999)
    sk.deskew(0.5 * tw, out)
    assert np.all(np.abs(out.download()["xyz"] - oracle.deskew(d0["xyz"], d0["t"], 0.5 * tw)) <= ulp)
    np.testing.assert_array_equal(sk.download()["xyz"], d0["xyz"])
    # skip_deskew / no time stamps: copy
    sk.deskew(None, out)
    np.testing.assert_array_equal(out.download()["xyz"], d0["xyz"])
    nots = capi.Scan(ctx, xyz)
    nots.deskew(tw, out)
    np.testing.assert_array_equal(out.download()["xyz"], xyz)


def test_deskew_pair_equals_separate_calls(ctx, small_workload):
    """mh_scan_deskew_pair: both layers + the bounding box of the de-skewed small one in one launch == mh_scan_deskew x 2 +
    mh_scan_bbox, bit for bit; the fallbacks (no twist, no time stamps, an empty layer) as well."""
    xyz, t = _raw(4, small_workload)
    raw = capi.Scan(ctx, xyz).set_timestamps(t)
    big, small = capi.Scan(ctx), capi.Scan(ctx)
    raw.preprocess(capi.preprocess_params(0.3, 1.2, min_points_to_filter=100, timestamp_method=capi.TS_MIDDLE_IS_ZERO), big, small)
    assert 0 < small.n < big.n
    tw = np.array([11.0, 0.4, -0.05, 0.02, -0.01, 0.5])
    for twist in (tw, None):
        a, b, a1, b1 = capi.Scan(ctx), capi.Scan(ctx), capi.Scan(ctx), capi.Scan(ctx)
        mn, mx, nf = big.deskew_pair(small, twist, a, b)
        big.deskew(twist, a1)
        small.deskew(twist, b1)
        for got, want in ((a, a1), (b, b1)):
            dg, dw = got.download(), want.download()
            for key in ("xyz", "t", "src_idx"):
                np.testing.assert_array_equal(dg[key], dw[key])
        mn1, mx1, nf1 = b1.bbox()
        np.testing.assert_array_equal(mn, mn1)
        np.testing.assert_array_equal(mx, mx1)
        assert nf == nf1 == small.n
    # a layer without time stamps / an empty small layer: the separate calls behind the same entry point
    nots_big, nots_small = capi.Scan(ctx, xyz[:5000]), capi.Scan(ctx, xyz[:300])
    a, b = capi.Scan(ctx), capi.Scan(ctx)
    mn, mx, nf = nots_big.deskew_pair(nots_small, tw, a, b)
    np.testing.assert_array_equal(b.download()["xyz"], xyz[:300])
    assert nf == np.isfinite(xyz[:300]).all(axis=1).sum()
    empty = capi.Scan(ctx)
    mn, mx, nf = big.deskew_pair(empty, tw, a, b)
    assert nf == 0 and b.n == 0 and a.n == big.n
    with pytest.raises(capi.MolahipError):
        big.deskew_pair(small, tw, a, a)


def _assert_maps_equal(g, o):
    for k in ("vox_keys", "vox_first", "vox_count", "src_idx", "xyz"):
        np.testing.assert_array_equal(g[k], o[k], err_msg=k)


@pytest.mark.parametrize("vs,cap,far", [(1.0, 20, 0.0), (1.0, 20, 45.0), (0.5, 4, 30.0)])
def test_map_insert_keyframes_bit_exact(ctx, oracle, vs, cap, far):
    """A drive of key-frames: every update (posed insertion after the stored content, cap, far-voxel removal)
    leaves exactly the voxel contents and source indices of the per-point CPU insertion.  The update is asynchronous
    (counts read back lazily by info() / download())."""
    scene = synth.make_scene(777, 80.0, 12)
    g, o = capi.Map(ctx, vs, cap), oracle.Map(vs, cap)
    offered = 0
    for k in range(6):
        pose = [-40.0 + 14.0 * k, 0.6 * np.sin(k), synth.SENSOR_H, 0.05 * k, 0.002 * k, -0.001 * k]
        xyz = synth.make_scan(scene, pose, rings=32, azimuths=400, seed=100 + k)
        T = synth.pose_from_ypr(pose)
        g.insert(capi.Scan(ctx, xyz), T, far)
        o.insert_posed(xyz, T, far)
        offered += len(xyz)
        i = g.info()
        assert (i.n_points, i.n_voxels, i.n_offered) == (o.num_points, o.num_voxels, offered)
        _assert_maps_equal(g.download(), o.dump())
        mn, mx = o.bbox()
        np.testing.assert_array_equal(np.array(i.bbox_min), mn)
        np.testing.assert_array_equal(np.array(i.bbox_max), mx)
    if far:
        assert o.num_points < oracle.Map(vs, cap).insert_posed(xyz, T).num_points * 6
    # the updated map answers NN queries like the oracle's
    q = capi.Scan(ctx, xyz)
    got = capi.nn_search(g, q, T, 1.5)
    ref = oracle.match_points(o, xyz, T, 1.5)
    np.testing.assert_array_equal(got["local_idx"], ref["local_idx"])
    np.testing.assert_array_equal(got["global_idx"], ref["global_idx"])
    np.testing.assert_array_equal(got["d2"], ref["d2"])


@pytest.mark.parametrize("metric", [1, 2])
@pytest.mark.parametrize("full_sort", ["collect", "three_launches", "full_sort"])
def test_map_insert_far_voxel_metric_switch(ctx, oracle, metric, full_sort, monkeypatch):
    """mh_map_params::far_voxel_metric (L1 / L2 readings of remove_voxels_farther_than, yaml:237-238) on the three insertion
    paths (merge with the fused collect kernel -- the default up to 0.4 M stored points --, merge with gather / compose /
    keys as separate launches, full sort), against the oracle."""
    if full_sort == "full_sort":
        monkeypatch.setenv("MH_MAP_FULL_SORT", "1")
    if full_sort == "three_launches":
        monkeypatch.setenv("MH_MAP_NO_COLLECT", "1")
    scene = synth.make_scene(779, 80.0, 12)
    g, o = capi.Map(ctx, 1.0, 20, far_voxel_metric=metric), oracle.Map(1.0, 20, far_voxel_metric=metric)
    cheb = oracle.Map(1.0, 20)
    for k in range(4):
        pose = [-30.0 + 16.0 * k, 0.5 * np.cos(k), synth.SENSOR_H, 0.04 * k, 0.0, 0.001 * k]
        xyz = synth.make_scan(scene, pose, rings=32, azimuths=400, seed=300 + k)
        T = synth.pose_from_ypr(pose)
        g.insert(capi.Scan(ctx, xyz), T, 40.0)
        o.insert_posed(xyz, T, 40.0)
        cheb.insert_posed(xyz, T, 40.0)
        _assert_maps_equal(g.download(), o.dump())
    assert o.num_voxels < cheb.num_voxels


def test_map_insert_ndt_and_empty(ctx, oracle):
    scene = synth.make_scene(778, 60.0, 8)
    kw = dict(min_distance_between_points=0.1, ndt_max_eigen_ratio=0.03, ndt_min_points=4)
    g, o = capi.Map(ctx, 2.0, 12, 0, 0.1, 0.03, 4), oracle.Map(2.0, 12, 0, **kw)
    g.insert(capi.Scan(ctx), np.eye(4)[:3], 10.0)  # empty key-frame into an empty map
    assert g.info().n_points == 0
    for k in range(3):
        pose = [-10.0 + 9.0 * k, 0.3 * k, synth.SENSOR_H, 0.1 * k, 0.0, 0.0]
        xyz = synth.make_scan(scene, pose, rings=32, azimuths=300, seed=200 + k)
        T = synth.pose_from_ypr(pose)
        g.insert(capi.Scan(ctx, xyz), T, 35.0)
        o.insert_posed(xyz, T, 35.0)
        _assert_maps_equal(g.download(), o.dump())
        gn, on = g.download_ndt(), o.dump_ndt()
        np.testing.assert_array_equal(gn["is_plane"], on["is_plane"])
        np.testing.assert_allclose(gn["centroid"], on["centroid"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(gn["normal"], on["normal"], rtol=0, atol=1e-5)
    assert g.info().n_planes > 10
    g.insert(capi.Scan(ctx), T, 35.0)  # empty key-frame: nothing changes
    _assert_maps_equal(g.download(), o.dump())


@pytest.mark.parametrize("lds", [None, "5", "70"])
@pytest.mark.parametrize("cap", [0, 150])
def test_min_distance_walk_by_waves(ctx, oracle, lds, cap, monkeypatch):
    """k_keep_seq (insertPoint's min_distance_between_points, lidar3d-ndt.yaml:244) walks a voxel run with a wave: runs longer
    than 64 entries, stored points in front of new ones, a cap reached in the middle of a run, more accepted points than the
    wave's LDS list holds (MH_KEEP_LDS shrinks the list: the overflow is found through the verdicts in memory) -- bit-equal to
    the oracle's sequential insertPoint on a full build and on key-frame insertions."""
    if lds:
        monkeypatch.setenv("MH_KEEP_LDS", lds)
    rng = np.random.default_rng(4242)
    vs, md = 4.0, 0.11
    g, o = capi.Map(ctx, vs, cap, 0, md), oracle.Map(vs, cap, 0, min_distance_between_points=md)
    # dense blobs inside a few voxels (hundreds of candidates per run, up to ~600 accepted) + scattered points
    def cloud(n_blob, seed):
        r = np.random.default_rng(seed)
        blobs = [c + r.uniform(-1.9, 1.9, (n_blob, 3)) for c in ([2.0, 2.0, 2.0], [-6.0, 2.0, 2.0], [10.0, -6.0, 2.0])]
        return np.concatenate(blobs + [r.uniform(-20, 20, (3000, 3))]).astype(np.float32)
    first = cloud(2500, 1)
    g.build(first)
    o.insert(first)
    _assert_maps_equal(g.download(), o.dump())
    assert o.num_points > 600
    I = np.eye(4)[:3]
    for k in range(3):
        xyz = cloud(1500, 10 + k)
        rng.shuffle(xyz)
        g.insert(capi.Scan(ctx, xyz), I, 1000.0)
        o.insert_posed(xyz, I, 1000.0)
        _assert_maps_equal(g.download(), o.dump())


def test_interleaved_upload_equals_channel_upload(ctx, small_workload):
    """mh_scan_update_aos: KITTI-style [n,4] rows, a PointCloud2-style record with the fields in odd places and a time
    stamp field, an empty buffer, and the argument checks."""
    xyz, t = _raw(3, small_workload)
    n = len(xyz)
    ref = capi.Scan(ctx, xyz).download()
    kitti = np.concatenate([xyz, np.full((n, 1), 0.5, np.float32)], 1)
    s = capi.Scan(ctx)
    s.update_interleaved(kitti)
    got = s.download()
    assert got["xyz"].tobytes() == ref["xyz"].tobytes() and len(s) == n
    # record: [intensity, z, t, x, ring, y]  (24 bytes)
    rec = np.zeros((n, 6), np.float32)
    rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3], rec[:, 4], rec[:, 5] = 7.0, xyz[:, 2], t, xyz[:, 0], 3.0, xyz[:, 1]
    s.update_interleaved(rec, off_x=12, off_y=20, off_z=4, off_t=8)
    got = s.download()
    assert got["xyz"].tobytes() == ref["xyz"].tobytes() and got["t"].tobytes() == t.tobytes()
    # ... and it feeds the filters like any other scan
    a, b = capi.Scan(ctx), capi.Scan(ctx)
    s.preprocess(capi.preprocess_params(timestamp_method=capi.TS_MIDDLE_IS_ZERO, **PP), a, None)
    capi.Scan(ctx, xyz).set_timestamps(t).preprocess(capi.preprocess_params(timestamp_method=capi.TS_MIDDLE_IS_ZERO, **PP), b, None)
    da, db = a.download(), b.download()
    assert da["xyz"].tobytes() == db["xyz"].tobytes() and da["t"].tobytes() == db["t"].tobytes()
    import torch  # device-resident records (MH_MEM_DEVICE): read in place
    s.update_interleaved(torch.from_numpy(rec).cuda(), off_x=12, off_y=20, off_z=4, off_t=8)
    got = s.download()
    assert got["xyz"].tobytes() == ref["xyz"].tobytes() and got["t"].tobytes() == t.tobytes()
    s.update_interleaved(kitti)  # no time stamp field: the channel is gone again
    assert not s.download()["t"].any()
    s.update_interleaved(np.zeros((0, 4), np.float32))
    assert len(s) == 0
    with pytest.raises(capi.MolahipError):
        s.update_interleaved(kitti, off_x=2)
    with pytest.raises(capi.MolahipError):
        s.update_interleaved(kitti, off_t=16)
