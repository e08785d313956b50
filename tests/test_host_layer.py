"""C++ host layer (namespace mp2p_icp_hip): the mirror of the mp2p_icp plugin API that sits on the C ABI.

CPU part: YAML-subset reader, ${ENV|default}, run-time formulas, pipeline construction by class name -- when
/root/reference is present (this container only) the reference's own pipelines/lidar3d-default.yaml is fed in
unchanged.  GPU part: ICP::align() driven exactly like LidarOdometry.cpp:961-962 and compared with the oracle."""
import os

import numpy as np
import pytest

from mola_lidar_odometry_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_YAML = "/root/reference/pipelines/lidar3d-default.yaml"
OUR_YAML = os.path.join(ROOT, "pipelines", "lidar3d-default-hip.yaml")
REF_NDT_YAML = "/root/reference/pipelines/lidar3d-ndt.yaml"
OUR_NDT_YAML = os.path.join(ROOT, "pipelines", "lidar3d-ndt-hip.yaml")


@pytest.fixture(scope="module")
def hl():
    import mola_lidar_odometry_amd.capi as capi
    capi.lib()  # load libmolahip (and torch's HIP runtime first) before the extension resolves its symbols
    from mola_lidar_odometry_amd import _mp2p_icp_hip
    return _mp2p_icp_hip


def test_expression_evaluator(hl):
    ev = hl.evaluate_expression
    assert ev("1+2*3", {}) == 7 and ev("-(2+3)^2", {}) == -25 and ev("max(1, 2, 3) - min(4, 5)", {}) == -1
    assert ev("2.0*max(S, 2.0*S-(2.0*S-0.5*S)*K/30)", {"S": 2.0, "K": 10.0}) == pytest.approx(6.0)
    assert ev("abs(-3)+sqrt(16)+clamp(5,0,2)", {}) == 9
    with pytest.raises(RuntimeError):
        ev("1+unknown_var", {})
    with pytest.raises(RuntimeError):
        ev("(1+2", {})


def test_yaml_subset_and_env_substitution(hl, monkeypatch):
    monkeypatch.setenv("MY_ITERS", "77")
    c = hl.Config.FromYamlText("""
a:
  b: ${MY_ITERS|300}   # comment
  c: ${NOT_SET_ANYWHERE|false}
  d: '$f{max(0.5, min(1.0, 0.015*40))}'
seq:
  - class: X
    params:
      t: 'x # not a comment'
  - {global: "g", local: "l", weight: 1.0}
""")
    assert c["a"]["b"].asString() == "77" and c["a"]["c"].asString() == "false"
    assert float(c["a"]["d"].asString()) == pytest.approx(0.6)
    assert c["seq"].size() == 2 and c["seq"].at(0)["class"].asString() == "X"
    assert c["seq"].at(0)["params"]["t"].asString() == "x # not a comment"
    assert c["seq"].at(1)["local"].asString() == "l"


@pytest.mark.parametrize("path", [OUR_YAML, REF_YAML])
def test_pipeline_from_yaml_builds_by_class_name(hl, path):
    if not os.path.exists(path):
        pytest.skip("reference tree not present on this box")
    cfg = hl.Config.FromYamlFile(path)["icp_settings_with_vel"]
    icp, params = hl.icp_pipeline_from_yaml(cfg)
    assert params.maxIterations == 300 and params.minAbsStep_trans == 1e-4 and params.minAbsStep_rot == 5e-5
    # the two run-time formulas of the pipeline (yaml:190,198) evaluate to the known schedule
    m = cfg["matchers"].at(0)["params"]
    s = cfg["solvers"].at(0)["params"]
    thr, kp = synth.threshold_schedule(2.0, 40)
    for k in (0, 7, 20, 39):
        v = {"ADAPTIVE_THRESHOLD_SIGMA": 2.0, "ICP_ITERATION": float(k)}
        assert hl.evaluate_expression(m["threshold"].asString(), v) == pytest.approx(thr[k])
        assert hl.evaluate_expression(s["robustKernelParam"].asString(), v) == pytest.approx(kp[k])
    assert m["pointLayerMatches"].at(0)["global"].asString() == "localmap"
    assert m["pointLayerMatches"].at(0)["local"].asString() == "decimated_for_icp"


@pytest.mark.parametrize("path", [OUR_NDT_YAML, REF_NDT_YAML])
def test_ndt_pipeline_from_yaml(hl, path):
    """lidar3d-ndt.yaml:162-215: Matcher_Point2Plane + Matcher_Points_DistanceThreshold, one GN step, 5e-4 stall steps."""
    if not os.path.exists(path):
        pytest.skip("reference tree not present on this box")
    cfg = hl.Config.FromYamlFile(path)["icp_settings_with_vel"]
    icp, params = hl.icp_pipeline_from_yaml(cfg)
    assert params.maxIterations == 300 and params.minAbsStep_trans == 5e-4 and params.minAbsStep_rot == 5e-4
    assert cfg["matchers"].size() == 2
    assert cfg["matchers"].at(0)["class"].asString().endswith("Matcher_Point2Plane")
    assert hl.evaluate_expression(cfg["matchers"].at(0)["params"]["distanceThreshold"].asString(),
                                  {"ADAPTIVE_THRESHOLD_SIGMA": 0.7}) == pytest.approx(0.7)
    assert cfg["solvers"].at(0)["params"]["maxIterations"].asString() == "1"


def test_unknown_class_is_an_error(hl):
    cfg = hl.Config.FromYamlText("""
class_name: mp2p_icp::ICP
solvers:
  - class: mp2p_icp::Solver_Horn
    params:
      ~
matchers:
  - class: mp2p_icp::Matcher_Points_DistanceThreshold
    params:
      threshold: 1.0
""")
    with pytest.raises(RuntimeError):
        hl.icp_pipeline_from_yaml(cfg)


# ------------------------------------------------------------------------------------------ GPU
def _maps(hl, w):
    g = hl.metric_map_t()
    hv = hl.HashedVoxelPointCloud(w.voxel_size, w.cap)
    hv.setPoints(w.map_xyz)
    g.set_layer("localmap", hv)
    l = hl.metric_map_t()
    l.set_layer("decimated_for_icp", hl.PointCloud(w.scan_xyz))
    return l, g, hv


@pytest.mark.gpu
@pytest.mark.parametrize("generic", [False, True])
def test_icp_align_through_the_plugin_api_matches_oracle(hl, oracle, small_workload, generic):
    """The call of LidarOdometry.cpp:961-962: icp->align(local, global, guess, params, result[, prior]) with the
    pipeline built from YAML and ADAPTIVE_THRESHOLD_SIGMA fed through a ParameterSource (LidarOdometry.cpp:1601-1604)."""
    w = small_workload
    cfg = hl.Config.FromYamlFile(OUR_YAML)["icp_settings_with_vel"]
    icp, params = hl.icp_pipeline_from_yaml(cfg)
    src = hl.ParameterSource()
    src.updateVariable("ADAPTIVE_THRESHOLD_SIGMA", w.sigma)
    src.updateVariable("ICP_ITERATION", 0)
    icp.attachToParameterSource(src)
    src.realize()
    icp.forceGenericPath(generic)
    l, g, hv = _maps(hl, w)
    assert hv.size() == len(w.map_xyz)
    res = icp.align(l, g, hl.TPose3D(*w.guess_ypr), params)
    assert icp.lastAlignUsedFusedPath() == (not generic)
    thr, kp = synth.threshold_schedule(w.sigma, 300)
    om = oracle.Map(w.voxel_size, w.cap).insert(w.map_xyz)
    o = oracle.icp_align(om, w.scan_xyz, w.T_guess, oracle.ICPParams(max_iterations=300, threshold=thr, kernel_param=kp),
                         want_pairs=True)
    assert res.nIterations == o["n_iterations"]
    assert res.terminationReason.name == oracle.TERM_NAMES[o["termination_reason"]]
    np.testing.assert_allclose(res.pose(), o["T"], atol=1e-7)
    assert res.quality == o["quality"] and res.n_pairs() == o["n_final_pairs"]
    np.testing.assert_array_equal(res.pair_global_idx(), o["pairs"]["global_idx"])
    np.testing.assert_allclose(np.reshape(res.cov(), (6, 6)), o["cov"], rtol=2e-5, atol=1e-6 * np.abs(o["cov"]).max())


@pytest.mark.gpu
def test_twist_hook_protocol_device_and_host_hooks_agree(hl, oracle, small_workload):
    """The in-tree hook (LidarOdometry.cpp:923-952) as a device-side test and as an arbitrary host callback must stop
    at the same iteration with the same pose; then the caller re-runs with the remaining budget (:956-967)."""
    w = small_workload
    cfg = hl.Config.FromYamlFile(OUR_YAML)["icp_settings_with_vel"]
    l, g, _ = _maps(hl, w)
    guess = hl.TPose3D(*w.guess_ypr)
    results = []
    for mode in ("device", "host", "replay"):
        icp, params = hl.icp_pipeline_from_yaml(cfg)
        src = hl.ParameterSource()
        src.updateVariable("ADAPTIVE_THRESHOLD_SIGMA", w.sigma)
        icp.attachToParameterSource(src)
        params.maxIterations = 40
        chk = hl.CPose3D(guess)
        if mode == "device":
            icp.setDeviceHook(0.15, float(np.deg2rad(0.75)), chk)
        else:
            def hook(it, T, chk=chk):
                d = hl.CPose3D.from_matrix(T) - chk
                t = np.asarray(d.matrix()).reshape(3, 4)
                ang = np.arccos(np.clip((np.trace(t[:, :3]) - 1) / 2, -1, 1))
                return bool(np.linalg.norm(t[:, 3]) > 0.15 or ang > np.deg2rad(0.75))
            icp.setIterationHook(hook)
            # "replay": the opaque hook on the FUSED loop (molahip_host/hook_replay.h, the mp2p_icp adapter's way):
            # trace, replay, re-run with the budget at which the hook asked to stop
            icp.setHookReplay(mode == "replay")
        results.append(icp.align(l, g, guess, params))
        assert icp.lastAlignUsedFusedPath() == (mode != "host")
    a, b, c = results
    assert a.terminationReason.name == b.terminationReason.name == c.terminationReason.name == "HookRequest"
    assert a.nIterations == b.nIterations == c.nIterations
    np.testing.assert_allclose(a.pose(), b.pose(), atol=1e-9)
    np.testing.assert_array_equal(a.pose(), c.pose())  # the same device loop, stopped by budget instead of by the device hook
    assert a.quality == c.quality and a.n_pairs() == c.n_pairs()
    np.testing.assert_array_equal(np.asarray(a.cov()), np.asarray(c.cov()))
    thr, kp = synth.threshold_schedule(w.sigma, 40)
    om = oracle.Map(w.voxel_size, w.cap).insert(w.map_xyz)
    o = oracle.icp_align(om, w.scan_xyz, w.T_guess, oracle.ICPParams(
        max_iterations=40, threshold=thr, kernel_param=kp, hook_enabled=True, hook_min_trans=0.15,
        hook_min_rot=float(np.deg2rad(0.75))))
    assert a.nIterations == o["n_iterations"]
    np.testing.assert_allclose(a.pose(), o["T"], atol=1e-7)


@pytest.mark.gpu
def test_failures_surface_as_exceptions(hl, small_workload):
    """The reference catches std::exception around the whole scan (LidarOdometry.cpp:614-619)."""
    w = small_workload
    cfg = hl.Config.FromYamlFile(OUR_YAML)["icp_settings_with_vel"]
    icp, params = hl.icp_pipeline_from_yaml(cfg)
    l, g, _ = _maps(hl, w)
    with pytest.raises(RuntimeError):  # ADAPTIVE_THRESHOLD_SIGMA was never published
        icp.align(l, g, hl.TPose3D(*w.guess_ypr), params)
    empty = hl.metric_map_t()
    src = hl.ParameterSource()
    src.updateVariable("ADAPTIVE_THRESHOLD_SIGMA", 2.0)
    icp.attachToParameterSource(src)
    with pytest.raises(RuntimeError):  # layer missing
        icp.align(empty, g, hl.TPose3D(*w.guess_ypr), params)


@pytest.mark.gpu
@pytest.mark.parametrize("generic,skip", [(False, False), (True, False), (False, True), (True, True)])
def test_ndt_pipeline_align_matches_oracle(hl, oracle, generic, skip, monkeypatch):
    """lidar3d-ndt.yaml driven through the plugin API against an mp2p_icp_hip::NDT map (config 5 of BASELINE.json); `skip`:
    MOLA_HIP_MATCHED_POINTS=skip (U12) through the fused loop and through the matcher-granular loop's pairing bookkeeping."""
    monkeypatch.setenv("MOLA_HIP_MATCHED_POINTS", "skip" if skip else "again")
    hl.reload_plugin_switches()
    assert hl.library_matched_points() == (1 if skip else 0)  # (the host library's own cache of the switches, not the module's)
    rng = np.random.default_rng(21)
    ground = np.stack([rng.uniform(-10, 10, 20000), rng.uniform(-10, 10, 20000), rng.normal(0.3, 0.01, 20000)], 1)
    wall = np.stack([rng.uniform(-10, 10, 12000), rng.normal(5.4, 0.01, 12000), rng.uniform(0.5, 4, 12000)], 1)
    blob = rng.normal([3.5, -3.5, 1.5], 0.25, (4000, 3))
    pts = np.concatenate([ground, wall, blob]).astype(np.float32)
    scan = pts[rng.permutation(len(pts))[:4000]]
    cfg = hl.Config.FromYamlFile(OUR_NDT_YAML)["icp_settings_with_vel"]
    icp, params = hl.icp_pipeline_from_yaml(cfg)
    src = hl.ParameterSource()
    sigma = 0.5
    src.updateVariable("ADAPTIVE_THRESHOLD_SIGMA", sigma)
    icp.attachToParameterSource(src)
    icp.forceGenericPath(generic)
    params.maxIterations = 60
    g = hl.metric_map_t()
    ndt = hl.NDT(1.0, 0, 0.2, 0.05)
    ndt.setPoints(pts)
    assert ndt.planeCount() > 100
    g.set_layer("localmap", ndt)
    l = hl.metric_map_t()
    l.set_layer("decimated_for_icp", hl.PointCloud(scan))
    guess_ypr = [0.10, -0.08, 0.05, 0.008, -0.004, 0.005]
    res = icp.align(l, g, hl.TPose3D(*guess_ypr), params)
    assert icp.lastAlignUsedFusedPath() == (not generic)
    om = oracle.Map(1.0, 0, 0, 0.2, 0.05, 4).insert(pts)
    thr, kp = synth.threshold_schedule(sigma, 60)
    o = oracle.icp_align(om, scan, oracle.pose_from_ypr(guess_ypr), oracle.ICPParams(
        max_iterations=60, min_abs_step_trans=5e-4, min_abs_step_rot=5e-4, threshold=thr, kernel_param=kp,
        pt2pl_threshold=1.0 * sigma, gn=oracle.GNParams(max_inner_iterations=1), pt2pt_skip_plane_paired=skip))
    monkeypatch.delenv("MOLA_HIP_MATCHED_POINTS")
    hl.reload_plugin_switches()
    assert res.nIterations == o["n_iterations"]
    assert res.terminationReason.name == oracle.TERM_NAMES[o["termination_reason"]]
    np.testing.assert_allclose(res.pose(), o["T"], atol=1e-7)
    assert res.n_pairs() == o["n_final_pairs"] and res.n_pairs_pt2pl() == o["n_final_pairs_pt2pl"] > 500
    assert res.quality == o["quality"]


def test_compiled_formulas_equal_the_interpreter():
    """ICP::align sweeps ICP_ITERATION over compiled formulas; they must give what the text interpreter gives."""
    from mola_lidar_odometry_amd import _mp2p_icp_hip as H
    v = {"ADAPTIVE_THRESHOLD_SIGMA": 1.37, "ICP_ITERATION": 7.0, "ESTIMATED_SENSOR_MAX_RANGE": 83.2, "wx": 0.01, "wy": -0.2,
         "wz": 0.4, "INSTANTANEOUS_SENSOR_MAX_RANGE": 71.0}
    exprs = ["2.0*max(ADAPTIVE_THRESHOLD_SIGMA, 2.0*ADAPTIVE_THRESHOLD_SIGMA-(2.0*ADAPTIVE_THRESHOLD_SIGMA-0.5*ADAPTIVE_THRESHOLD_SIGMA)*ICP_ITERATION/30)",
             "(0.1e-2 + sqrt(wx^2+wy^2+wz^2)*0.1)*ESTIMATED_SENSOR_MAX_RANGE", "(15 + sqrt(wx^2+wy^2+wz^2)*500 )",
             "max(0.20, 0.55*1e-2*ESTIMATED_SENSOR_MAX_RANGE)", "-0.20*INSTANTANEOUS_SENSOR_MAX_RANGE", "-(1+2)*-3^2",
             "clamp(ICP_ITERATION/2, 1, 3) + min(4, 5, wx) - abs(wy) + pow(2, 3) + pi", "1.0*ADAPTIVE_THRESHOLD_SIGMA"]
    for e in exprs:
        assert H.evaluate_compiled(e, v) == H.evaluate_expression(e, v), e
    for bad in ("max()", "foo(1)", "1 +", "nope + 1"):
        with pytest.raises(RuntimeError):
            H.evaluate_compiled(bad, v)


@pytest.mark.gpu
def test_generate_debug_files_writes_the_iteration_trace(hl, oracle, small_workload, tmp_path):
    """MP2P_ICP_GENERATE_DEBUG_FILES (lidar3d-default.yaml:177-182): one trace file per align(); its content is the
    oracle's per-iteration trajectory."""
    import json
    w = small_workload
    icp, params = hl.icp_pipeline_from_yaml(hl.Config.FromYamlFile(OUR_YAML)["icp_settings_with_vel"])
    src = hl.ParameterSource()
    src.updateVariable("ADAPTIVE_THRESHOLD_SIGMA", w.sigma)
    src.updateVariable("ICP_ITERATION", 0)
    icp.attachToParameterSource(src)
    src.realize()
    params.generateDebugFiles = True
    params.debugFileNameFormat = str(tmp_path / "logs" / "icp-run-$UNIQUE_ID-local_$LOCAL_ID$LOCAL_LABEL.icplog.json")
    l, g, _ = _maps(hl, w)
    res = icp.align(l, g, hl.TPose3D(*w.guess_ypr), params)
    files = sorted(os.listdir(tmp_path / "logs"))
    assert len(files) == 1 and files[0].endswith(".icplog.json") and "$" not in files[0]
    log = json.load(open(tmp_path / "logs" / files[0]))
    thr, kp = synth.threshold_schedule(w.sigma, 300)
    o = oracle.icp_align(oracle.Map(w.voxel_size, w.cap).insert(w.map_xyz), w.scan_xyz, w.T_guess,
                         oracle.ICPParams(max_iterations=300, threshold=thr, kernel_param=kp))
    assert log["format"] == "molahip-icplog-json-1" and log["n_iterations"] == o["n_iterations"] == res.nIterations
    assert log["termination"] == oracle.TERM_NAMES[o["termination_reason"]] and log["n_local_points"] == len(w.scan_xyz)
    assert [it["n_pairs"] for it in log["iterations"]] == [t["n_pairs"] for t in o["trace"]]
    np.testing.assert_allclose([it["pose"] for it in log["iterations"]], [t["T"] for t in o["trace"]], atol=1e-9)
    np.testing.assert_allclose([it["threshold"] for it in log["iterations"]], thr[:len(log["iterations"])])
    np.testing.assert_allclose(log["final_pose"], o["T"], atol=1e-9)
    icp.align(l, g, hl.TPose3D(*w.guess_ypr), params)
    assert len(os.listdir(tmp_path / "logs")) == 2


# ------------------------------------------------------------------------------------------ matcher generality (a7)
_NEAR_FAR_ICP = """
class_name: mp2p_icp::ICP
params:
  maxIterations: 60
  minAbsStep_trans: 1e-4
  minAbsStep_rot: 5e-5
solvers:
  - class: mp2p_icp::Solver_GaussNewton
    params:
      maxIterations: 2
      robustKernel: 'RobustKernel::GemanMcClure'
      robustKernelParam: '0.5*max(ADAPTIVE_THRESHOLD_SIGMA, 2.0*ADAPTIVE_THRESHOLD_SIGMA-(2.0*ADAPTIVE_THRESHOLD_SIGMA-0.5*ADAPTIVE_THRESHOLD_SIGMA)*ICP_ITERATION/30)'
matchers:
  - class: mp2p_icp::Matcher_Points_DistanceThreshold
    params:
      threshold: '2.0*max(ADAPTIVE_THRESHOLD_SIGMA, 2.0*ADAPTIVE_THRESHOLD_SIGMA-(2.0*ADAPTIVE_THRESHOLD_SIGMA-0.5*ADAPTIVE_THRESHOLD_SIGMA)*ICP_ITERATION/30)'
      thresholdAngularDeg: 0
      pairingsPerPoint: 1
      allowMatchAlreadyMatchedGlobalPoints: true
      runFromIteration: 4
      runUpToIteration: 0
      pointLayerMatches:
        - {global: "localmap_far", local: "decimated_for_icp_far", weight: 1.0}
  - class: mp2p_icp::Matcher_Points_DistanceThreshold
    params:
      threshold: '2.00*ADAPTIVE_THRESHOLD_SIGMA'
      thresholdAngularDeg: 0
      pairingsPerPoint: 1
      allowMatchAlreadyMatchedGlobalPoints: true
      runFromIteration: 0
      runUpToIteration: %d
      pointLayerMatches:
        - {global: "localmap_near", local: "decimated_for_icp_near", weight: 1.0}
        - {global: "localmap_far", local: "decimated_for_icp_near", weight: 1.0}
quality:
  - class: mp2p_icp::QualityEvaluator_PairedRatio
    params:
      ~
"""


@pytest.mark.gpu
@pytest.mark.parametrize("up_to", [0, 9])
def test_gated_matchers_on_several_layers_match_an_oracle_loop(hl, oracle, small_workload, up_to):
    """The ICP block shape of the reference's pipelines/extras/lidar3d-near-far.yaml:150-199 (two point matchers, an iteration
    gate `runFromIteration: 4`, two pointLayerMatches entries on the second one; here also a `runUpToIteration`) is NOT one of
    the fused shapes: it runs matcher by matcher on the device (mh_nn_search per layer pair, mh_gn_solve per iteration -- the
    path the adapter's Matcher_*_HIP / Solver_GaussNewton_HIP classes take on a real stack).  Checked against the same loop
    written with the oracle's matcher and solver."""
    w = small_workload
    sigma = 1.2
    scan = w.scan_xyz
    rng = np.linalg.norm(scan, axis=1)
    near_l, far_l = scan[rng < 9.0], scan[rng >= 6.0]           # overlapping on purpose
    T0 = w.T_gt.reshape(3, 4)
    mp = w.map_xyz
    mr = np.linalg.norm(mp - T0[:, 3], axis=1)
    near_g, far_g = mp[mr < 12.0], mp[::2]
    g = hl.metric_map_t()
    for name, pts, vs in (("localmap_near", near_g, 0.5), ("localmap_far", far_g, 1.0)):
        hv = hl.HashedVoxelPointCloud(vs, 20)
        hv.setPoints(pts)
        g.set_layer(name, hv)
    l = hl.metric_map_t()
    l.set_layer("decimated_for_icp_near", hl.PointCloud(near_l))
    l.set_layer("decimated_for_icp_far", hl.PointCloud(far_l))
    icp, params = hl.icp_pipeline_from_yaml(hl.Config.FromYamlText(_NEAR_FAR_ICP % up_to))
    src = hl.ParameterSource()
    src.updateVariable("ADAPTIVE_THRESHOLD_SIGMA", sigma)
    src.updateVariable("ICP_ITERATION", 0)
    icp.attachToParameterSource(src)
    src.realize()
    res = icp.align(l, g, hl.TPose3D(*w.guess_ypr), params)
    assert not icp.lastAlignUsedFusedPath()

    # the same loop on the oracle
    maps = {"localmap_near": oracle.Map(0.5, 20).insert(near_g), "localmap_far": oracle.Map(1.0, 20).insert(far_g)}
    locs = {"decimated_for_icp_near": near_l, "decimated_for_icp_far": far_l}
    base = lambda k: max(sigma, 2.0 * sigma - (2.0 * sigma - 0.5 * sigma) * k / 30.0)  # noqa: E731
    matchers = [dict(thr=lambda k: 2.0 * base(k), run_from=4, up_to=0, layers=[("localmap_far", "decimated_for_icp_far")]),
                dict(thr=lambda k: 2.0 * sigma, run_from=0, up_to=up_to,
                     layers=[("localmap_near", "decimated_for_icp_near"), ("localmap_far", "decimated_for_icp_near")])]
    T, Tprev, term, it, n_pairs, potential = w.T_guess.copy(), w.T_guess.copy(), "MaxIterations", 0, 0, 0
    for it in range(60):
        lp, gp, potential = [], [], 0
        for m in matchers:
            if (m["run_from"] and it < m["run_from"]) or (m["up_to"] and it > m["up_to"]):
                continue
            for gname, lname in m["layers"]:
                r = oracle.match_points(maps[gname], locs[lname], T, m["thr"](it))
                lp.append(locs[lname][r["local_idx"]])
                gp.append(r["global_xyz"] if "global_xyz" in r else np.stack([r["gx"], r["gy"], r["gz"]], 1))
                potential += len(locs[lname])
        n_pairs = sum(len(a) for a in lp)
        if n_pairs == 0:
            term = "NoPairings"
            break
        T = oracle.gn_solve(T, pt2pt=(np.concatenate(lp), np.concatenate(gp)),
                            params=oracle.GNParams(max_inner_iterations=2, robust_kernel_param=0.5 * base(it)))[0]
        d = oracle.se3_log(oracle.pose_compose(oracle.pose_inverse(Tprev), T))
        if np.linalg.norm(d[:3]) < 1e-4 and np.linalg.norm(d[3:]) < 5e-5:
            term = "Stalled"
            break
        Tprev = T.copy()
    else:
        it = 60
    assert res.terminationReason.name == term
    assert res.nIterations == it
    np.testing.assert_allclose(res.pose(), T, atol=1e-7)
    assert res.n_pairs() == n_pairs and res.quality == pytest.approx(n_pairs / potential, abs=1e-12)
    assert np.abs(res.pose() - w.T_gt).max() < 0.05  # and it converged to the ground truth


_TWO_PAIRINGS_ICP = """
class_name: mp2p_icp::ICP
params:
  maxIterations: 40
  minAbsStep_trans: 1e-4
  minAbsStep_rot: 5e-5
solvers:
  - class: mp2p_icp::Solver_GaussNewton
    params:
      maxIterations: 2
      robustKernel: 'RobustKernel::GemanMcClure'
      robustKernelParam: 0.3
matchers:
  - class: mp2p_icp::Matcher_Points_DistanceThreshold
    params:
      threshold: 0.9
      thresholdAngularDeg: 0.5
      pairingsPerPoint: 2
      allowMatchAlreadyMatchedGlobalPoints: true
      pointLayerMatches:
        - {global: "localmap", local: "decimated_for_icp", weight: 1.0}
quality:
  - class: mp2p_icp::QualityEvaluator_PairedRatio
    params:
      ~
"""


@pytest.mark.gpu
def test_two_pairings_per_point_match_an_oracle_loop(hl, oracle, small_workload):
    """`pairingsPerPoint: 2` with an angular threshold (the point matcher of the reference's pipelines/rgbd.yaml:133-141): the
    matcher runs nn_multiple_search on the device (mh_nn_search_k), every local point contributes up to two pairings to the
    Gauss-Newton step and two to potential_pairings.  Checked against the same loop written with the oracle's k-best matcher and
    solver."""
    w = small_workload
    g = hl.metric_map_t()
    hv = hl.HashedVoxelPointCloud(1.0, 20)
    hv.setPoints(w.map_xyz)
    g.set_layer("localmap", hv)
    l = hl.metric_map_t()
    l.set_layer("decimated_for_icp", hl.PointCloud(w.scan_xyz))
    icp, params = hl.icp_pipeline_from_yaml(hl.Config.FromYamlText(_TWO_PAIRINGS_ICP))
    res = icp.align(l, g, hl.TPose3D(*w.guess_ypr), params)
    assert not icp.lastAlignUsedFusedPath()

    om = oracle.Map(1.0, 20).insert(w.map_xyz)
    T, Tprev, term, it, n_pairs = w.T_guess.copy(), w.T_guess.copy(), "MaxIterations", 0, 0
    for it in range(40):
        r = oracle.match_points_k(om, w.scan_xyz, T, 0.9, 2, 0.5)
        n_pairs = len(r["local_idx"])
        if n_pairs == 0:
            term = "NoPairings"
            break
        T = oracle.gn_solve(T, pt2pt=(w.scan_xyz[r["local_idx"]], r["global_xyz"]),
                            params=oracle.GNParams(max_inner_iterations=2, robust_kernel_param=0.3))[0]
        d = oracle.se3_log(oracle.pose_compose(oracle.pose_inverse(Tprev), T))
        if np.linalg.norm(d[:3]) < 1e-4 and np.linalg.norm(d[3:]) < 5e-5:
            term = "Stalled"
            break
        Tprev = T.copy()
    else:
        it = 40
    assert n_pairs > len(w.scan_xyz)  # most points have two partners
    assert res.terminationReason.name == term
    assert res.nIterations == it
    np.testing.assert_allclose(res.pose(), T, atol=1e-7)
    assert res.n_pairs() == n_pairs and res.quality == pytest.approx(n_pairs / (2 * len(w.scan_xyz)), abs=1e-12)


_RGBD_ICP = """
class_name: mp2p_icp::ICP
params:
  maxIterations: 30
  minAbsStep_trans: 1e-4
  minAbsStep_rot: 5e-5
solvers:
  - class: mp2p_icp::Solver_GaussNewton
    params:
      maxIterations: 2
      robustKernel: 'RobustKernel::GemanMcClure'
      robustKernelParam: 0.3
matchers:
  - class: mp2p_icp::Matcher_Points_DistanceThreshold
    params:
      threshold: 0.9
      thresholdAngularDeg: 0.5
      pairingsPerPoint: 2
      allowMatchAlreadyMatchedGlobalPoints: true
      pointLayerMatches:
        - {global: "localmap", local: "decimated_for_icp", weight: 1.0}
  - class: mp2p_icp::Matcher_Point2Plane
    params:
      distanceThreshold: 0.40
      planeEigenThreshold: 1e-2
      searchRadius: 0.80
      knn: 10
      minimumPlanePoints: 6
      pointLayerMatches:
        - {global: "localmap", local: "decimated_for_icp", weight: 1.0}
quality:
  - class: mp2p_icp::QualityEvaluator_PairedRatio
    params:
      ~
"""


@pytest.mark.gpu
def test_rgbd_shaped_icp_block_point_pairs_and_knn_pca_planes_match_an_oracle_loop(hl, oracle, small_workload):
    """The ICP block shape of the reference's pipelines/rgbd.yaml:118-151 -- Matcher_Points_DistanceThreshold with two pairings
    per point, then Matcher_Point2Plane on a plain HashedVoxelPointCloud layer (k nearest neighbours + PCA: mh_nn_search_pt2pl_knn,
    round 5) -- through the plugin-API mirror, against the same loop written with the oracle's matchers and solver."""
    w = small_workload
    l, g, _ = _maps(hl, w)
    icp, params = hl.icp_pipeline_from_yaml(hl.Config.FromYamlText(_RGBD_ICP))
    res = icp.align(l, g, hl.TPose3D(*w.guess_ypr), params)
    assert not icp.lastAlignUsedFusedPath()

    om = oracle.Map(w.voxel_size, w.cap).insert(w.map_xyz)
    T, Tprev, term, it, n_pairs, n_pl = w.T_guess.copy(), w.T_guess.copy(), "MaxIterations", 0, 0, 0
    for it in range(30):
        r = oracle.match_points_k(om, w.scan_xyz, T, 0.9, 2, 0.5)
        q = oracle.match_pt2pl_knn(om, w.scan_xyz, T, 0.40, 1e-2, 0.80, 10, 6)
        n_pl = len(q["local_idx"])
        n_pairs = len(r["local_idx"]) + n_pl
        if n_pairs == 0:
            term = "NoPairings"
            break
        T = oracle.gn_solve(T, pt2pt=(w.scan_xyz[r["local_idx"]], r["global_xyz"]),
                            pt2pl=(w.scan_xyz[q["local_idx"]], q["centroid"], q["normal"]),
                            params=oracle.GNParams(max_inner_iterations=2, robust_kernel_param=0.3))[0]
        d = oracle.se3_log(oracle.pose_compose(oracle.pose_inverse(Tprev), T))
        if np.linalg.norm(d[:3]) < 1e-4 and np.linalg.norm(d[3:]) < 5e-5:
            term = "Stalled"
            break
        Tprev = T.copy()
    else:
        it = 30
    assert n_pl > 100  # the street canyon's ground and facades give planes
    assert res.terminationReason.name == term
    assert res.nIterations == it
    np.testing.assert_allclose(res.pose(), T, atol=1e-7)
    assert res.n_pairs() == n_pairs


@pytest.mark.gpu
@pytest.mark.parametrize("generic", [False, True])
def test_layer_weight_other_than_one_is_a_device_input(hl, oracle, small_workload, generic):
    """`pointLayerMatches: - {global, local, weight: w}` (lidar3d-default.yaml:203-204; Pairings::point_weights [U]) with w != 1:
    the weight scales the point pairs' rows against the prior factor, so it changes the pose only when a prior is given -- the
    fused loop and the matcher / solver loop both pass it to the device solver (round 5; rounds 1-4 fell back or ignored it),
    checked against the oracle with the same weight and against the weight-1 run (which must differ)."""
    w = small_workload
    yaml = open(OUR_YAML).read()
    assert "weight: 1.0" in yaml
    poses = {}
    info = (np.eye(6) * np.array([4e4, 4e4, 4e4, 1e5, 1e5, 1e5])).reshape(36).tolist()
    Tp = w.T_gt.copy()
    Tp[3] += 0.20  # a prior that pulls 20 cm along x
    for weight in (1.0, 3.0):
        cfg = hl.Config.FromYamlText(yaml.replace("weight: 1.0", "weight: %.1f" % weight))["icp_settings_with_vel"]
        icp, params = hl.icp_pipeline_from_yaml(cfg)
        src = hl.ParameterSource()
        src.updateVariable("ADAPTIVE_THRESHOLD_SIGMA", w.sigma)
        src.updateVariable("ICP_ITERATION", 0)
        icp.attachToParameterSource(src)
        src.realize()
        icp.forceGenericPath(generic)
        l, g, _ = _maps(hl, w)
        prior = hl.CPose3DPDFGaussianInf(hl.CPose3D.from_matrix(Tp.tolist()), info)
        res = icp.align(l, g, hl.TPose3D(*w.guess_ypr), params, prior)
        assert icp.lastAlignUsedFusedPath() == (not generic)
        thr, kp = synth.threshold_schedule(w.sigma, 300)
        om = oracle.Map(w.voxel_size, w.cap).insert(w.map_xyz)
        o = oracle.icp_align(om, w.scan_xyz, w.T_guess, oracle.ICPParams(max_iterations=300, threshold=thr, kernel_param=kp,
                             gn=oracle.GNParams(weight_pt2pt=weight)), prior=(Tp, np.reshape(info, (6, 6))))
        assert res.nIterations == o["n_iterations"]
        np.testing.assert_allclose(res.pose(), o["T"], atol=1e-7)
        poses[weight] = np.asarray(res.pose())
    assert np.abs(poses[1.0] - poses[3.0]).max() > 1e-4  # the weight matters against a prior
