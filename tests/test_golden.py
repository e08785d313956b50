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
