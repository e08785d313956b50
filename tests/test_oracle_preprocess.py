"""CPU: the oracle's restatement of the observation filters and of the incremental local-map
update (row f2), pinned against brute-force numpy restatements written independently here."""
import numpy as np
import pytest

from mola_lidar_odometry_amd import synth


def _cloud(seed, n=30000):
    rng = np.random.default_rng(seed)
    xyz = np.concatenate([rng.normal(0, 1, (n, 3)) * [35, 35, 2.5], rng.uniform(-1, 1, (n // 10, 3))]).astype(np.float32)
    t = rng.uniform(1000.0, 1000.1, len(xyz)).astype(np.float32)
    return xyz, t


def _brute_first_point(xyz, res, mode=0):
    s = xyz * (np.float32(1.0) / np.float32(res))
    k = (np.trunc(s) if mode else np.floor(s)).astype(np.int64)
    _, first = np.unique(k, axis=0, return_index=True)
    return np.sort(first).astype(np.uint32)


@pytest.mark.parametrize("res,mode", [(0.2, 0), (0.55, 0), (1.6, 0), (0.6, 1)])
def test_decimate_first_point(oracle, res, mode):
    xyz, _ = _cloud(1)
    np.testing.assert_array_equal(oracle.decimate_first_point(xyz, res, 2000, mode), _brute_first_point(xyz, res, mode))


def test_decimate_passthrough_and_nonfinite(oracle):
    xyz, _ = _cloud(2, 1500)
    xyz[5] = [np.nan, 0, 0]
    idx = oracle.decimate_first_point(xyz, 0.5, 2000)  # smaller than minimum_input_points_to_filter
    np.testing.assert_array_equal(idx, np.delete(np.arange(len(xyz)), 5))
    assert len(oracle.decimate_first_point(xyz, 0.5, 100)) < len(xyz) - 1
    assert len(oracle.decimate_first_point(np.zeros((0, 3), np.float32), 0.5, 0)) == 0


def _brute_closest_to_average(xyz, res, mode=0):
    """An independent reading of DecimateMethod::ClosestToAverage: per voxel, float32 running sums in input order, mean =
    sum * (1 / count) in float32, squared error (dx*dx + dy*dy) + dz*dz in float32, argmin keeping the first of equals."""
    f = np.float32
    s = xyz * (f(1.0) / f(res))
    k = (np.trunc(s) if mode else np.floor(s)).astype(np.int64)
    _, inv = np.unique(k, axis=0, return_inverse=True)
    inv = np.asarray(inv).reshape(-1)
    keep = []
    for v in range(inv.max() + 1):
        members = np.flatnonzero(inv == v)  # ascending = input order
        sums = np.zeros(3, f)
        for i in members:
            sums = (sums + xyz[i]).astype(f)
        mean = (sums * (f(1.0) / f(len(members)))).astype(f)
        d = (xyz[members] - mean).astype(f)
        e = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(f) + d[:, 2] * d[:, 2]).astype(f)
        keep.append(members[int(np.argmin(e))])  # (argmin returns the first minimum)
    return np.sort(np.array(keep)).astype(np.uint32)


@pytest.mark.parametrize("res,mode", [(0.4, 0), (1.6, 0), (0.9, 1)])
def test_decimate_closest_to_average(oracle, res, mode):
    xyz, _ = _cloud(3, 6000)
    xyz[::7] = np.round(xyz[::7] * 4) / 4  # points on a lattice: exact ties between candidates of a voxel
    got = oracle.decimate_closest_to_average(xyz, res, 100, mode)
    np.testing.assert_array_equal(got, _brute_closest_to_average(xyz, res, mode))
    assert len(got) == len(oracle.decimate_first_point(xyz, res, 100, mode))  # one survivor per voxel either way
    assert not np.array_equal(got, oracle.decimate_first_point(xyz, res, 100, mode))


def test_decimate_closest_to_average_passthrough_nonfinite_and_chain(oracle):
    xyz, _ = _cloud(4, 1500)
    xyz[5] = [np.nan, 0, 0]
    np.testing.assert_array_equal(oracle.decimate_closest_to_average(xyz, 0.5, 2000), np.delete(np.arange(len(xyz)), 5))
    assert len(oracle.decimate_closest_to_average(np.zeros((0, 3), np.float32), 0.5, 0)) == 0
    # the chain with the method switched per stage = the stages one by one
    xyz, _ = _cloud(5, 8000)
    for mm, mi in ((1, 0), (0, 1), (1, 1)):
        im, ii = oracle.preprocess(xyz, 0.4, 1.2, 100, decim_map_method=mm, decim_icp_method=mi)
        a = (oracle.decimate_closest_to_average if mm else oracle.decimate_first_point)(xyz, 0.4, 100)
        np.testing.assert_array_equal(im, a)
        b = (oracle.decimate_closest_to_average if mi else oracle.decimate_first_point)(xyz[a], 1.2, 100)
        np.testing.assert_array_equal(ii, a[b])


def test_range_and_bbox(oracle):
    xyz, _ = _cloud(3)
    c = np.array([0.5, -0.25, 0.1], np.float32)
    d = xyz - c
    sq = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    rmin, rmax = np.float32(3.0), np.float32(60.0)
    np.testing.assert_array_equal(oracle.filter_by_range(xyz, rmin, rmax, c),
                                  np.nonzero((sq >= rmin * rmin) & (sq <= rmax * rmax))[0])
    mn, mx = np.array([-20, -20, 0.1], np.float32), np.array([20, 20, 1.0], np.float32)
    inside = np.all((xyz >= mn) & (xyz <= mx), axis=1)
    np.testing.assert_array_equal(oracle.filter_bbox(xyz, mn, mx, True), np.nonzero(inside)[0])
    np.testing.assert_array_equal(oracle.filter_bbox(xyz, mn, mx, False), np.nonzero(~inside)[0])


def test_adjust_timestamps(oracle):
    _, t = _cloud(4)
    mid = np.float32(0.5) * (t.min() + t.max())
    np.testing.assert_array_equal(oracle.adjust_timestamps(t, oracle.TS_MIDDLE_IS_ZERO, 0.25), (t - mid) + np.float32(0.25))
    np.testing.assert_array_equal(oracle.adjust_timestamps(t, oracle.TS_EARLIEST_IS_ZERO), t - t.min())
    np.testing.assert_array_equal(oracle.adjust_timestamps(t, oracle.TS_NONE), t)


def test_deskew_against_matrix_exponential(oracle):
    from scipy.linalg import expm
    xyz, t = _cloud(5, 300)
    t = oracle.adjust_timestamps(t)
    tw = np.array([12.0, -0.4, 0.1, 0.02, -0.03, 0.6])
    got = oracle.deskew(xyz, t, tw)
    W = np.array([[0, -tw[5], tw[4]], [tw[5], 0, -tw[3]], [-tw[4], tw[3], 0]])
    for i in range(len(xyz)):
        dt = float(t[i])
        ref = expm(W * dt) @ xyz[i].astype(np.float64) + tw[:3] * dt
        np.testing.assert_allclose(got[i], ref, rtol=0, atol=1e-5)
    np.testing.assert_array_equal(oracle.deskew(xyz, t, np.zeros(6)), xyz)  # zero twist: identity


def test_preprocess_chain_equals_stage_by_stage(oracle):
    xyz, _ = _cloud(6)
    kw = dict(range_min=1.0, range_max=90.0, bbox_mode=1, bbox_min=(-15, -15, 0.5), bbox_max=(15, 15, 7.0))
    im, ii = oracle.preprocess(xyz, 0.4, 1.2, 2000, **kw)
    a = _brute_first_point(xyz, 0.4)
    a = a[oracle.filter_by_range(xyz[a], 1.0, 90.0)]
    a = a[oracle.filter_bbox(xyz[a], kw["bbox_min"], kw["bbox_max"], False)]
    np.testing.assert_array_equal(im, a)
    np.testing.assert_array_equal(ii, a[_brute_first_point(xyz[a], 1.2)])
    assert 0 < len(ii) < len(im) < len(xyz)


def test_incremental_insert_equals_one_shot_and_eviction(oracle):
    rng = np.random.default_rng(7)
    a = (rng.normal(0, 1, (20000, 3)) * [25, 25, 2]).astype(np.float32)
    b = (rng.normal(0, 1, (15000, 3)) * [25, 25, 2]).astype(np.float32)
    I = np.eye(4)[:3]
    inc = oracle.Map(1.0, 5).insert_posed(a, I).insert_posed(b, I)
    one = oracle.Map(1.0, 5).insert(np.concatenate([a, b]))
    for k, v in one.dump().items():
        np.testing.assert_array_equal(inc.dump()[k], v)
    # posed insertion = insertion of the composed points
    T = synth.pose_from_ypr([3.0, -2.0, 0.5, 0.3, 0.02, -0.01]).reshape(3, 4)
    bg = ((T[:, :3] @ b.astype(np.float64).T).T + T[:, 3]).astype(np.float32)
    p1 = oracle.Map(1.0, 5).insert_posed(b, T).dump()
    p2 = oracle.Map(1.0, 5).insert(bg).dump()
    np.testing.assert_array_equal(p1["vox_keys"], p2["vox_keys"])
    np.testing.assert_allclose(p1["xyz"], p2["xyz"], rtol=0, atol=4e-6)  # numpy's matmul sums in another order
    # far-voxel removal: Chebyshev distance in voxel units from the voxel of the insertion pose
    T2 = np.eye(4)[:3].copy()
    T2[:, 3] = [30.2, -4.7, 0.3]
    ev = oracle.Map(1.0, 5).insert_posed(a, I).insert_posed(b, T2, 20.0)
    d = ev.dump()
    c = np.floor(np.float32(T2[:, 3])).astype(np.int64)
    assert np.abs(d["vox_keys"] - c).max() == 20
    full = oracle.Map(1.0, 5).insert_posed(a, I).insert_posed(b, T2, 0.0).dump()
    keep_v = np.abs(full["vox_keys"] - c).max(axis=1) <= 20
    np.testing.assert_array_equal(d["vox_keys"], full["vox_keys"][keep_v])
    keep_p = np.repeat(keep_v, full["vox_count"])
    np.testing.assert_array_equal(d["xyz"], full["xyz"][keep_p])
    np.testing.assert_array_equal(d["src_idx"], full["src_idx"][keep_p])
    assert ev.num_points == keep_p.sum()
    mn, mx = ev.bbox()
    np.testing.assert_array_equal(mn, d["xyz"].min(0))
    np.testing.assert_array_equal(mx, d["xyz"].max(0))
    # an evicted voxel can be re-populated from empty
    ev.insert_posed(a, I, 0.0)
    assert ev.num_points > keep_p.sum()


@pytest.mark.parametrize("metric", [1, 2])
def test_far_voxel_metric_switch(oracle, metric):
    """remove_voxels_farther_than with the other readings of "farther" (lidar3d-default.yaml:237 says L1; upstream's code is
    unverified): brute force over the dumped voxel keys."""
    rng = np.random.default_rng(17)
    a = (rng.normal(0, 1, (30000, 3)) * [25, 25, 4]).astype(np.float32)
    I = np.eye(4)[:3]
    T2 = I.copy()
    T2[:, 3] = [6.2, -4.7, 0.3]
    full = oracle.Map(1.0, 5, far_voxel_metric=metric).insert_posed(a, I).dump()
    ev = oracle.Map(1.0, 5, far_voxel_metric=metric).insert_posed(a, I).insert_posed(a[:0], T2, 20.0).dump()
    dk = np.abs(full["vox_keys"].astype(np.int64) - np.floor(np.float32(T2[:, 3])).astype(np.int64))
    keep = dk.sum(1) <= 20 if metric == 1 else (dk * dk).sum(1) <= 400
    assert 0 < keep.sum() < len(keep)
    np.testing.assert_array_equal(ev["vox_keys"], full["vox_keys"][keep])
    np.testing.assert_array_equal(ev["xyz"], full["xyz"][np.repeat(keep, full["vox_count"])])
    cheb = oracle.Map(1.0, 5).insert_posed(a, I).insert_posed(a[:0], T2, 20.0)
    assert cheb.num_voxels > len(ev["vox_keys"])  # the default (Chebyshev) keeps the corners the others drop
