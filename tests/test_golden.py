"""Committed golden vectors (tests/golden/small_align.json, generator: tests/golden/make_golden.py): the C oracle must
reproduce them on CPU; the HIP path must reproduce them on the GPU.  They pin the restatement (both CPU oracles agree
on them), not the reference -- see DESIGN.md section 5."""
import json
import os

import numpy as np
import pytest

from mola_lidar_odometry_amd import synth

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "small_align.json")))


def test_inputs_are_the_committed_ones(small_workload):
    w = small_workload
    assert (len(w.scan_xyz), len(w.map_xyz)) == (GOLD["n_scan"], GOLD["n_map"])
    assert float(w.scan_xyz.astype(np.float64).sum()) == GOLD["scan_checksum"]
    assert float(w.map_xyz.astype(np.float64).sum()) == GOLD["map_checksum"]
    np.testing.assert_array_equal(w.T_guess, GOLD["T_guess"])


def _check(run_fixed, run_stall, names):
    f = run_fixed()
    g = GOLD["fixed20"]
    assert [t["n_pairs"] for t in f["trace"]] == g["n_pairs_per_iteration"]
    np.testing.assert_allclose([t["T"] for t in f["trace"]], g["T_per_iteration"], atol=1e-9)
    np.testing.assert_allclose(f["T"], g["T_final"], atol=1e-9)
    assert f["quality"] == g["quality"]
    np.testing.assert_allclose(np.diag(f["cov"]), g["cov_diag"], rtol=2e-5)
    s = run_stall()
    gs = GOLD["stall300"]
    assert s["n_iterations"] == gs["n_iterations"] and names[s["termination_reason"]] == gs["termination"]
    assert s["n_final_pairs"] == gs["n_final_pairs"]
    np.testing.assert_allclose(s["T"], gs["T_final"], atol=1e-9)


def test_c_oracle_reproduces_golden(oracle, small_workload):
    w = small_workload
    om = oracle.Map(w.voxel_size, w.cap).insert(w.map_xyz)
    thr, kp = synth.threshold_schedule(w.sigma, 300)
    _check(lambda: oracle.icp_align(om, w.scan_xyz, w.T_guess, oracle.ICPParams(
               max_iterations=w.n_iters, disable_stall_test=True, threshold=w.threshold, kernel_param=w.kernel_param)),
           lambda: oracle.icp_align(om, w.scan_xyz, w.T_guess, oracle.ICPParams(max_iterations=300, threshold=thr,
                                                                                 kernel_param=kp)),
           oracle.TERM_NAMES)


def test_numpy_oracle_rows_agree_with_c_oracle(oracle, small_workload):
    """The numpy rows of the fixture (independent float64 restatement) against the C oracle on the same subsample."""
    w = small_workload
    om = oracle.Map(w.voxel_size, w.cap).insert(w.map_xyz)
    g = GOLD["numpy_subsample_every8_4iters"]
    r = oracle.icp_align(om, w.scan_xyz[::8], w.T_guess, oracle.ICPParams(
        max_iterations=4, disable_stall_test=True, threshold=w.threshold[:4], kernel_param=w.kernel_param[:4]))
    assert [t["n_pairs"] for t in r["trace"]] == g["n_pairs_per_iteration"]
    np.testing.assert_allclose([t["T"] for t in r["trace"]], g["T_per_iteration"], atol=1e-9)


@pytest.mark.gpu
def test_hip_path_reproduces_golden(small_workload):
    from mola_lidar_odometry_amd import capi
    w = small_workload
    ctx = capi.Context(0)
    gm = capi.Map(ctx, w.voxel_size, w.cap).build(w.map_xyz)
    gs = capi.Scan(ctx, w.scan_xyz)
    thr, kp = synth.threshold_schedule(w.sigma, 300)
    _check(lambda: capi.icp_align(gm, gs, w.T_guess, capi.ICPParams(
               max_iterations=w.n_iters, disable_stall_test=True, threshold=w.threshold, kernel_param=w.kernel_param)),
           lambda: capi.icp_align(gm, gs, w.T_guess, capi.ICPParams(max_iterations=300, threshold=thr, kernel_param=kp)),
           capi.TERM_NAMES)
    ctx.close()


# ------------------------------------------------------------------------------------------------ rows f1-f3
FRONT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "frontend.json")))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _crc(a):
    import zlib
    return int(zlib.crc32(np.ascontiguousarray(a).tobytes()))


@pytest.fixture(scope="module")
def front_drive():
    d = synth.make_drive(FRONT["drive"]["n_scans"])
    xyz, t = d["scans"][5]
    assert (len(xyz), _crc(xyz), _crc(t)) == (FRONT["drive"]["scan5_points"], FRONT["drive"]["scan5_crc"], FRONT["drive"]["scan5_t_crc"])
    return d


def _check_frontend(pre, deskew, map_dump, recs):
    g = FRONT["preprocess"]
    im, ii = pre
    assert (len(im), len(ii), _crc(im), _crc(ii)) == (g["n_map"], g["n_icp"], g["idx_map_crc"], g["idx_icp_crc"])
    assert [int(v) for v in ii[:8]] == g["idx_icp_head"] and [int(v) for v in ii[-8:]] == g["idx_icp_tail"]
    if deskew is not None:  # (floating point: the device may differ from libm by one float ulp on a few coordinates)
        np.testing.assert_allclose(deskew[:3], FRONT["deskew"]["first3"], rtol=0, atol=4e-6)
    gm = FRONT["map_insert"]
    assert (len(map_dump["xyz"]), len(map_dump["vox_keys"])) == (gm["n_points"], gm["n_voxels"])
    assert (_crc(map_dump["xyz"]), _crc(map_dump["src_idx"]), _crc(map_dump["vox_keys"]), _crc(map_dump["vox_count"])) == (
        gm["xyz_crc"], gm["src_crc"], gm["keys_crc"], gm["count_crc"])
    go = FRONT["odometry"]
    for key in ("icp_iterations", "twist_corrections", "map_updated", "n_for_icp", "n_map_points"):
        assert [type(go[key][0])(r[key]) for r in recs] == go[key], key
    np.testing.assert_allclose([np.asarray(r["pose"]) for r in recs], go["poses"], rtol=0, atol=1e-6)
    np.testing.assert_allclose([r["sigma"] for r in recs], go["sigma"], rtol=0, atol=1e-8)


def test_c_oracle_reproduces_frontend_golden(oracle, front_drive):
    from oracle import odometry_oracle as oo
    d = front_drive
    xyz, t = d["scans"][5]
    pre = oracle.preprocess(xyz, **FRONT["preprocess"]["params"])
    ta = oracle.adjust_timestamps(t, oracle.TS_MIDDLE_IS_ZERO, 0.0)
    dsk = oracle.deskew(xyz[pre[1]], ta[pre[1]], FRONT["deskew"]["twist"])
    assert _crc(dsk) == FRONT["deskew"]["xyz_crc_of_icp_layer"]
    m = oracle.Map(1.0, 20)
    for k in FRONT["map_insert"]["keyframes"]:
        m.insert_posed(d["scans"][k][0][::3], d["poses"][k], FRONT["map_insert"]["remove_voxels_farther_than"])
    o = oo.OdometryOracle(os.path.join(ROOT, FRONT["odometry"]["pipeline"]), n_threads=8)
    recs = [o.on_lidar(st, s[0], s[1]) for s, st in zip(d["scans"], d["stamps"])]
    _check_frontend(pre, dsk, m.dump(), recs)


@pytest.mark.gpu
def test_hip_path_reproduces_frontend_golden(front_drive):
    from mola_lidar_odometry_amd import _mp2p_icp_hip as H
    from mola_lidar_odometry_amd import capi
    d = front_drive
    ctx = capi.Context(0)
    xyz, t = d["scans"][5]
    raw = capi.Scan(ctx, xyz).set_timestamps(t)
    om, oi, out = capi.Scan(ctx), capi.Scan(ctx), capi.Scan(ctx)
    raw.preprocess(capi.preprocess_params(timestamp_method=capi.TS_MIDDLE_IS_ZERO, **FRONT["preprocess"]["params"]), om, oi)
    oi.deskew(FRONT["deskew"]["twist"], out)
    m = capi.Map(ctx, 1.0, 20)
    for k in FRONT["map_insert"]["keyframes"]:
        m.insert(capi.Scan(ctx, d["scans"][k][0][::3]), d["poses"][k], FRONT["map_insert"]["remove_voxels_farther_than"])
    lo = H.LidarOdometry()
    lo.initialize(H.Config.FromYamlFile(os.path.join(ROOT, FRONT["odometry"]["pipeline"])))
    recs = [lo.onLidar(st, s[0], s[1]) for s, st in zip(d["scans"], d["stamps"])]
    _check_frontend((om.download()["src_idx"], oi.download()["src_idx"]), out.download()["xyz"], m.download(), recs)
    ctx.close()


# ------------------------------------------------------------------------------------------------ NDT / point-to-plane
NDT = json.load(open(os.path.join(ROOT, "tests", "golden", "ndt.json")))


def _ndt_inputs():
    pts = synth.ndt_cloud(NDT["cloud"]["seed"])
    assert (len(pts), _crc(pts)) == (NDT["cloud"]["n"], NDT["cloud"]["crc"])
    rng = np.random.default_rng(12)
    scan = pts[rng.permutation(len(pts))[:NDT["align"]["n_scan"]]]
    thr, kp = synth.threshold_schedule(0.5, 60)
    kw = dict(max_iterations=60, min_abs_step_trans=5e-4, min_abs_step_rot=5e-4, threshold=thr, kernel_param=kp,
              pt2pl_threshold=0.5)
    return pts, scan, np.asarray(NDT["align"]["guess"]), kw


def _check_ndt(info, ndt_dump, al, term_name, recs):
    g = NDT["map"]
    assert (info[0], info[1], int(ndt_dump["is_plane"].sum())) == (g["n_points"], g["n_voxels"], g["n_planes"])
    assert (_crc(ndt_dump["is_plane"].astype(np.uint32)), _crc(ndt_dump["centroid"])) == (g["plane_crc"], g["centroid_crc"])
    a = NDT["align"]
    assert (int(al["n_iterations"]), term_name, int(al["n_final_pairs"]), int(al["n_final_pairs_pt2pl"])) == (
        a["n_iterations"], a["termination"], a["n_final_pairs"], a["n_final_pairs_pt2pl"])
    np.testing.assert_allclose(np.asarray(al["T"]).reshape(-1), a["T_final"], rtol=0, atol=1e-9)
    d = NDT["drive"]
    for key in ("icp_iterations", "n_for_icp", "n_map_points", "map_updated"):
        assert [type(d[key][0])(r[key]) for r in recs] == d[key], key
    np.testing.assert_allclose([np.asarray(r["pose"]).reshape(-1) for r in recs], d["poses"], rtol=0, atol=1e-6)


def test_c_oracle_reproduces_ndt_golden(oracle):
    from oracle import odometry_oracle as oo
    pts, scan, guess, kw = _ndt_inputs()
    m = oracle.Map(1.0, 0, 0, 0.1, 0.05, 4).insert(pts)
    al = oracle.icp_align(m, scan, guess, oracle.ICPParams(gn=oracle.GNParams(max_inner_iterations=2), **kw))
    d = synth.make_drive(NDT["drive"]["n_scans"])
    o = oo.OdometryOracle(os.path.join(ROOT, NDT["drive"]["pipeline"]), n_threads=8)
    recs = [o.on_lidar(st, s[0], s[1]) for s, st in zip(d["scans"], d["stamps"])]
    _check_ndt((m.num_points, m.num_voxels), m.dump_ndt(), al, oracle.TERM_NAMES[al["termination_reason"]], recs)


@pytest.mark.gpu
def test_hip_path_reproduces_ndt_golden():
    from mola_lidar_odometry_amd import _mp2p_icp_hip as H
    from mola_lidar_odometry_amd import capi
    from oracle import oracle_c
    pts, scan, guess, kw = _ndt_inputs()
    ctx = capi.Context(0)
    m = capi.Map(ctx, 1.0, 0, 0, 0.1, 0.05, 4).build(pts)
    al = capi.icp_align(m, capi.Scan(ctx, scan), guess, capi.ICPParams(gn=capi.GNParams(max_inner_iterations=2), **kw))
    d = synth.make_drive(NDT["drive"]["n_scans"])
    lo = H.LidarOdometry()
    lo.initialize(H.Config.FromYamlFile(os.path.join(ROOT, NDT["drive"]["pipeline"])))
    recs = [lo.onLidar(st, s[0], s[1]) for s, st in zip(d["scans"], d["stamps"])]
    info = m.info()
    _check_ndt((int(info.n_points), int(info.n_voxels)), m.download_ndt(), al, oracle_c.TERM_NAMES[al["termination_reason"]], recs)
    ctx.close()
