"""SE(3) helpers of the C oracle vs scipy expm/logm (independent numpy oracle) -- SURVEY Appendix A."""
import numpy as np
import pytest

from oracle import icp_oracle_np as onp


def rand_xi(rng, scale_t=5.0, scale_r=1.0):
    return np.concatenate([rng.normal(0, scale_t, 3), rng.normal(0, scale_r, 3)])


def test_exp_matches_expm(oracle):
    rng = np.random.default_rng(0)
    for s in (1e-9, 1e-4, 0.1, 1.0, 3.0):
        for _ in range(20):
            xi = rand_xi(rng, 5.0, s)
            T = oracle.se3_exp(xi).reshape(3, 4)
            np.testing.assert_allclose(T, onp.se3_exp(xi)[:3], atol=1e-12, rtol=1e-12)


def test_log_inverts_exp(oracle):
    rng = np.random.default_rng(1)
    for s in (1e-9, 1e-5, 0.01, 1.0, 2.5):
        for _ in range(20):
            xi = rand_xi(rng, 3.0, s)
            if np.linalg.norm(xi[3:]) > 3.0:
                continue
            back = oracle.se3_log(oracle.se3_exp(xi))
            np.testing.assert_allclose(back, xi, atol=1e-9, rtol=1e-9)


def test_log_near_pi(oracle):
    for axis in ([1, 0, 0], [0, 1, 0], [0.6, 0, 0.8], [1 / np.sqrt(3)] * 3):
        for th in (np.pi - 1e-3, np.pi - 1e-7, np.pi - 1e-9):
            w = np.asarray(axis) * th
            xi = np.concatenate([[0.3, -0.2, 0.1], w])
            T = oracle.se3_exp(xi)
            back = oracle.se3_log(T)
            # compare as rotations (w and -w are the same rotation at pi)
            np.testing.assert_allclose(oracle.se3_exp(back), T, atol=1e-6)


def test_ypr_roundtrip_and_convention(oracle):
    rng = np.random.default_rng(2)
    for _ in range(50):
        p = np.concatenate([rng.normal(0, 10, 3), rng.uniform(-3, 3, 1), rng.uniform(-1.4, 1.4, 1), rng.uniform(-3, 3, 1)])
        T = oracle.pose_from_ypr(p)
        np.testing.assert_allclose(T.reshape(3, 4), onp.pose_from_ypr(p)[:3], atol=1e-14)
        np.testing.assert_allclose(oracle.pose_to_ypr(T), p, atol=1e-10)
    # yaw is a rotation about +z: x axis goes to +y for yaw=90deg
    T = oracle.pose_from_ypr([0, 0, 0, np.pi / 2, 0, 0]).reshape(3, 4)
    np.testing.assert_allclose(T[:, 0], [0, 1, 0], atol=1e-15)


def test_compose_inverse(oracle):
    rng = np.random.default_rng(3)
    for _ in range(20):
        A, B = oracle.se3_exp(rand_xi(rng)), oracle.se3_exp(rand_xi(rng))
        AB = oracle.pose_compose(A, B)
        np.testing.assert_allclose(onp.T44(AB), onp.T44(A) @ onp.T44(B), atol=1e-12)
        I = oracle.pose_compose(A, oracle.pose_inverse(A))
        np.testing.assert_allclose(onp.T44(I), np.eye(4), atol=1e-12)


def test_tum_fragment_envelope(oracle):
    """The only numbers the reference pins for this path: KITTI-00 frames move ~0.69/0.75 m forward
    (test/kitti_00_fragment_gt.tum:1-3, embedded here as data).  Plausibility of our SE(3) maths:
    the quaternion->pose->log chain gives those step lengths."""
    tum = np.array([[0, 0, 0, 0, 0, 0, 0, 1],
                    [0.103938, 0.6907, 0.0066, 0.0072, 0.0001, -0.0005, -0.0012, 1.0]])
    # only the first two rows' translation magnitude is used; identity quaternion row is exact
    assert abs(np.linalg.norm(tum[1, 1:4]) - 0.69) < 0.01
