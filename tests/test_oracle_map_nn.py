"""Voxel map + NN of the C oracle vs the independent numpy oracle and brute force."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import icp_oracle_np as onp


def test_floor_semantics_negative_coords(oracle):
    m = oracle.Map(1.0, 20)
    m.insert(np.array([[-0.2, 0.5, 0.5], [0.2, 0.5, 0.5], [-1.0, 0.0, -0.0]], np.float32))
    d = m.dump()
    assert d["vox_keys"].tolist() == [[-1, 0, 0], [0, 0, 0]]
    assert d["vox_count"].tolist() == [2, 1]
    mt = oracle.Map(1.0, 20, oracle.INDEX_TRUNC)
    mt.insert(np.array([[-0.2, 0.5, 0.5], [0.2, 0.5, 0.5]], np.float32))
    assert mt.num_voxels == 1  # truncation merges (-1,1) into voxel 0


def test_voxel_cap_keeps_first_in_order(oracle):
    rng = np.random.default_rng(0)
    pts = rng.uniform(0, 1, (100, 3)).astype(np.float32)
    m = oracle.Map(1.0, 20).insert(pts)
    assert m.num_points == 20 and m.num_voxels == 1
    d = m.dump()
    np.testing.assert_array_equal(d["src_idx"], np.arange(20))
    np.testing.assert_array_equal(d["xyz"], pts[:20])
    m0 = oracle.Map(1.0, 0).insert(pts)  # cap 0 = unlimited
    assert m0.num_points == 100


def test_nonfinite_points_dropped(oracle):
    pts = np.array([[0.5, 0.5, 0.5], [np.nan, 0, 0], [np.inf, 0, 0], [1.5, 0.5, 0.5]], np.float32)
    m = oracle.Map(1.0, 20).insert(pts)
    assert m.num_points == 2
    np.testing.assert_array_equal(sorted(m.dump()["src_idx"]), [0, 3])


def test_nn_reach_is_27_block_not_threshold(oracle):
    """SURVEY App.B U3: a map point 2.5 voxels away is invisible even with an 8 m threshold."""
    m = oracle.Map(1.0, 20).insert(np.array([[2.9, 0.5, 0.5]], np.float32))
    ok, _, _, _ = m.nn_single([0.5, 0.5, 0.5])
    assert not ok
    ok, pt, d2, idx = m.nn_single([1.1, 0.5, 0.5])
    assert ok and idx == 0 and abs(d2 - 1.8 ** 2) < 1e-5


def test_nn_tie_break_first_in_scan_order(oracle):
    # two points at the same distance in different voxels: x-outer order wins
    m = oracle.Map(1.0, 20).insert(np.array([[1.5, 0.5, 0.5], [-0.5, 0.5, 0.5]], np.float32))
    ok, pt, d2, idx = m.nn_single([0.5, 0.5, 0.5])
    assert ok and idx == 1  # voxel x=-1 is scanned before x=+1
    # same voxel: insertion order wins
    m = oracle.Map(1.0, 20).insert(np.array([[0.75, 0.5, 0.5], [0.25, 0.5, 0.5]], np.float32))
    ok, pt, d2, idx = m.nn_single([0.5, 0.5, 0.5])
    assert ok and idx == 0


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 10_000), vs=st.sampled_from([0.5, 1.0, 1.7]), cap=st.sampled_from([0, 3, 20]))
def test_nn_equals_bruteforce_on_27_block(oracle, seed, vs, cap):
    rng = np.random.default_rng(seed)
    pts = rng.normal(0, 3, (400, 3)).astype(np.float32)
    qs = rng.normal(0, 3, (60, 3)).astype(np.float32)
    mc = oracle.Map(vs, cap).insert(pts)
    mp = onp.VoxelMap(vs, cap).insert(pts)
    assert mc.num_points == mp.num_points
    for q in qs:
        ok, pt, d2, idx = mc.nn_single(q)
        r = mp.nn_single(q)
        assert ok == (r is not None)
        if ok:
            assert idx == r[0] and np.float32(d2) == r[2]
            np.testing.assert_array_equal(pt, r[1])


def test_match_points_vs_numpy(oracle, small_workload):
    w = small_workload
    mc = oracle.Map(w.voxel_size, w.cap).insert(w.map_xyz)
    mp = onp.VoxelMap(w.voxel_size, w.cap).insert(w.map_xyz)
    sub = w.scan_xyz[::4]
    for thr in (8.0, 0.7):
        a = oracle.match_points(mc, sub, w.T_guess, thr)
        b = onp.match_points(mp, sub, onp.T44(w.T_guess), thr)
        np.testing.assert_array_equal(a["local_idx"], b["local_idx"])
        np.testing.assert_array_equal(a["global_idx"], b["global_idx"])
        np.testing.assert_array_equal(a["d2"], b["d2"])
        np.testing.assert_array_equal(a["global_xyz"], b["global_xyz"])
        assert a["n_candidates"] == b["n_candidates"]
        assert a["potential_pairings"] == len(sub)
    assert len(a["local_idx"]) < len(sub)  # 0.7 m threshold rejects some


def test_match_threads_identical(oracle, small_workload):
    w = small_workload
    m = oracle.Map(w.voxel_size, w.cap).insert(w.map_xyz)
    a = oracle.match_points(m, w.scan_xyz, w.T_guess, 1.0, 0.0, 1)
    b = oracle.match_points(m, w.scan_xyz, w.T_guess, 1.0, 0.0, 4)
    for k in ("local_idx", "global_idx", "d2"):
        np.testing.assert_array_equal(a[k], b[k])


def test_empty_inputs(oracle):
    m = oracle.Map(1.0, 20)
    r = oracle.match_points(m, np.zeros((5, 3), np.float32), np.eye(4)[:3].reshape(12), 8.0)
    assert len(r["local_idx"]) == 0 and r["potential_pairings"] == 5
    m.insert(np.zeros((1, 3), np.float32))
    r = oracle.match_points(m, np.zeros((0, 3), np.float32), np.eye(4)[:3].reshape(12), 8.0)
    assert len(r["local_idx"]) == 0 and r["potential_pairings"] == 0
