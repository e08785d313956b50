"""Gauss-Newton solver / covariance of the C oracle vs the independent numpy oracle."""
import numpy as np
import pytest

from oracle import icp_oracle_np as onp


def make_pairs(rng, n=200, noise=0.05):
    l = rng.normal(0, 10, (n, 3)).astype(np.float32)
    Tt = onp.se3_exp(np.concatenate([rng.normal(0, 0.3, 3), rng.normal(0, 0.03, 3)]))
    q = (l.astype(np.float64) @ Tt[:3, :3].T + Tt[:3, 3] + rng.normal(0, noise, (n, 3))).astype(np.float32)
    return l, q, Tt


def make_planes(rng, n=150):
    l = rng.normal(0, 10, (n, 3)).astype(np.float32)
    nrm = rng.normal(0, 1, (n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    c = (l + rng.normal(0, 0.2, (n, 3))).astype(np.float32)
    return l, c, nrm.astype(np.float32)


@pytest.mark.parametrize("kernel", [0, 1, 2, 3, 4, 5])
def test_H_g_match_numpy_all_kernels(oracle, kernel):
    rng = np.random.default_rng(kernel)
    l, q, _ = make_pairs(rng)
    pl = make_planes(rng)
    T0 = onp.se3_exp(np.concatenate([rng.normal(0, 0.1, 3), rng.normal(0, 0.01, 3)]))
    p = oracle.GNParams(max_inner_iterations=2, robust_kernel=kernel, robust_kernel_param=0.7)
    T1, n, steps = oracle.gn_solve(onp.T12(T0), (l, q), pl, p)
    T_ref, steps_ref = onp.gn_solve(T0, (l, q), pl, 2, kernel, 0.7)
    assert n == 2
    for a, b in zip(steps, steps_ref):
        np.testing.assert_allclose(a["H"], b["H"], rtol=1e-11, atol=1e-9)
        np.testing.assert_allclose(a["g"], b["g"], rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(a["err_norm_sqr"], b["cost"], rtol=1e-12)
        np.testing.assert_allclose(a["delta"], b["delta"], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(onp.T44(T1), T_ref, atol=1e-10)


def test_gradient_is_right_perturbation(oracle):
    """g = J^T e must be the gradient of 1/2 sum|e|^2 wrt eps in T*exp(eps) (kernel none)."""
    rng = np.random.default_rng(7)
    l, q, _ = make_pairs(rng, 50)
    T0 = onp.se3_exp(np.array([0.2, -0.1, 0.05, 0.02, -0.01, 0.03]))
    p = oracle.GNParams(max_inner_iterations=1, robust_kernel=oracle.KERNEL_NONE)
    _, _, steps = oracle.gn_solve(onp.T12(T0), (l, q), None, p)

    def cost(eps):
        T = T0 @ onp.se3_exp(eps)
        e = l.astype(np.float64) @ T[:3, :3].T + T[:3, 3] - q
        return 0.5 * np.sum(e * e)

    h = 1e-6
    gfd = np.array([(cost(np.eye(6)[j] * h) - cost(-np.eye(6)[j] * h)) / (2 * h) for j in range(6)])
    np.testing.assert_allclose(steps[0]["g"], gfd, rtol=1e-6, atol=1e-6)


def test_exact_recovery_no_noise(oracle):
    rng = np.random.default_rng(3)
    l = rng.normal(0, 10, (100, 3)).astype(np.float32)
    Tt = onp.se3_exp(np.array([0.3, -0.2, 0.1, 0.01, 0.02, -0.015]))
    q64 = l.astype(np.float64) @ Tt[:3, :3].T + Tt[:3, 3]
    q = q64.astype(np.float32)
    p = oracle.GNParams(max_inner_iterations=6, robust_kernel=oracle.KERNEL_NONE)
    T1, n, _ = oracle.gn_solve(np.eye(4)[:3].reshape(12), (l, q), None, p)
    np.testing.assert_allclose(onp.T44(T1), Tt, atol=2e-6)  # limited by fp32 rounding of q


def test_identity_zero_residual_breaks_before_solve(oracle):
    l = np.random.default_rng(0).normal(0, 5, (30, 3)).astype(np.float32)
    p = oracle.GNParams(max_inner_iterations=2, robust_kernel=oracle.KERNEL_GM_C4, robust_kernel_param=1.0)
    T1, n, steps = oracle.gn_solve(np.eye(4)[:3].reshape(12), (l, l), None, p)
    assert n == 0  # errNorm <= maxCost(0): early exit (U8)
    np.testing.assert_array_equal(T1, np.eye(4)[:3].reshape(12))


def test_prior_term_matches_numpy_and_regularises_planar_wall(oracle):
    rng = np.random.default_rng(5)
    # planar wall z=0 with pt2pl only: H rank 3 (z, roll, pitch observable)
    l = np.stack([rng.uniform(-10, 10, 200), rng.uniform(-10, 10, 200), np.zeros(200) + 0.1], 1).astype(np.float32)
    c = np.stack([l[:, 0], l[:, 1], np.zeros(200)], 1).astype(np.float32)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (200, 1))
    p = oracle.GNParams(max_inner_iterations=1, robust_kernel=oracle.KERNEL_NONE)
    T0 = np.eye(4)
    _, _, steps = oracle.gn_solve(onp.T12(T0), None, (l, c, nrm), p)
    assert np.linalg.matrix_rank(steps[0]["H"], tol=1e-6) == 3
    # with a prior the system is full rank and matches numpy
    Tp = onp.se3_exp(np.array([0.05, -0.02, 0.0, 0, 0, 0.01]))
    Lam = np.diag([10, 10, 10, 100, 100, 100.0])
    T1, n, steps = oracle.gn_solve(onp.T12(T0), None, (l, c, nrm), p, prior=(onp.T12(Tp), Lam))
    T_ref, steps_ref = onp.gn_solve(T0, None, (l, c, nrm), 1, 0, 1.0, prior=(Tp, Lam))
    np.testing.assert_allclose(steps[0]["H"], steps_ref[0]["H"], rtol=1e-8, atol=1e-7)
    np.testing.assert_allclose(steps[0]["g"], steps_ref[0]["g"], rtol=1e-8, atol=1e-7)
    np.testing.assert_allclose(onp.T44(T1), T_ref, atol=1e-9)
    assert abs(T1[11]) < 0.11 and np.all(np.isfinite(T1))


def test_rank_deficient_without_prior_does_not_blow_up(oracle):
    l = np.stack([np.linspace(-5, 5, 50), np.linspace(3, -3, 50), np.full(50, 0.2)], 1).astype(np.float32)
    c = l.copy(); c[:, 2] = 0
    nrm = np.tile(np.array([0, 0, 1], np.float32), (50, 1))
    p = oracle.GNParams(max_inner_iterations=1, robust_kernel=oracle.KERNEL_NONE)
    T1, n, _ = oracle.gn_solve(np.eye(4)[:3].reshape(12), None, (l, c, nrm), p)
    assert n in (1, -1) and np.all(np.isfinite(T1))


def test_threads_equal_to_1e12(oracle):
    rng = np.random.default_rng(11)
    l, q, _ = make_pairs(rng, 5000)
    p = oracle.GNParams(max_inner_iterations=2, robust_kernel=1, robust_kernel_param=0.5)
    a, _, sa = oracle.gn_solve(np.eye(4)[:3].reshape(12), (l, q), None, p, n_threads=1)
    b, _, sb = oracle.gn_solve(np.eye(4)[:3].reshape(12), (l, q), None, p, n_threads=4)
    np.testing.assert_allclose(sa[0]["H"], sb[0]["H"], rtol=1e-12)
    np.testing.assert_allclose(a, b, atol=1e-12)


def test_covariance_matches_numpy(oracle):
    rng = np.random.default_rng(9)
    l, q, Tt = make_pairs(rng, 300)
    cov, ata = oracle.covariance(onp.T12(Tt), (l, q))
    ref = onp.covariance(Tt, (l, q))
    np.testing.assert_allclose(cov, ref, rtol=2e-5, atol=1e-12)
    assert np.all(np.linalg.eigvalsh(cov) > 0)
    cov0, _ = oracle.covariance(onp.T12(Tt), (np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32)))
    np.testing.assert_array_equal(cov0, np.eye(6) * 1e6)
