"""Independent float64 numpy/scipy restatement of the ICP hot path (second opinion for the C oracle).

TEST INFRASTRUCTURE (see oracle/icp_oracle.h).  PARITY UNPINNED: like the C oracle this follows
SURVEY.md 8(a) + Appendix A, not a runnable reference.  It deliberately shares NO code and as few
formulations as possible with icp_oracle.c:
  * SE(3) exp/log go through scipy.linalg.expm / logm of the 4x4 twist matrix;
  * the point-to-point Jacobian is built the MRPT way, (3x12 d(Rl+t)/d[d1 d2 d3 t]) x
    (12x6 dDexp(e)/de), instead of the closed form [R | -R[l]x];
  * the voxel map is a python dict of lists; the NN is a brute-force argmin over the 27-block;
  * the 6x6 system is solved with numpy.linalg.solve.
Used by tests/test_oracle_*.py to pin icp_oracle.c and by tests/golden/make_golden.py.
All "file:line" citations are relative to /root/reference.
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import expm, logm

# robust kernels -- same enum values as icp_oracle.h
KERNEL_NONE, KERNEL_GM_C4, KERNEL_GM_KISS, KERNEL_GM_BARRON, KERNEL_CAUCHY, KERNEL_GM_C2 = range(6)


def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)


def T44(T12):
    T = np.eye(4)
    T[:3, :] = np.asarray(T12, dtype=np.float64).reshape(3, 4)
    return T


def T12(T44_):
    return np.asarray(T44_)[:3, :].reshape(12).copy()


def se3_exp(xi):
    """exp([v;w]) via the 4x4 matrix exponential."""
    xi = np.asarray(xi, dtype=np.float64)
    M = np.zeros((4, 4))
    M[:3, :3] = hat(xi[3:])
    M[:3, 3] = xi[:3]
    return expm(M)


def se3_log(T):
    M = np.real(logm(np.asarray(T, dtype=np.float64)))
    return np.array([M[0, 3], M[1, 3], M[2, 3], M[2, 1], M[0, 2], M[1, 0]])


def pose_from_ypr(p):
    """TPose3D (x,y,z,yaw,pitch,roll) -> 4x4, R = Rz(yaw) Ry(pitch) Rx(roll)."""
    x, y, z, yaw, pitch, roll = p
    Rz = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(pitch), 0, np.sin(pitch)], [0, 1, 0], [-np.sin(pitch), 0, np.cos(pitch)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(roll), -np.sin(roll)], [0, np.sin(roll), np.cos(roll)]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = [x, y, z]
    return T


def robust_weight(kernel, c, e2):
    if kernel == KERNEL_GM_C4:
        return c ** 4 / (c * c + e2) ** 2
    if kernel == KERNEL_GM_KISS:
        return c * c / (c + e2) ** 2
    if kernel == KERNEL_GM_BARRON:
        return 1.0 / (e2 / (4 * c * c) + 1.0) ** 2
    if kernel == KERNEL_CAUCHY:
        return c * c / (c * c + e2)
    if kernel == KERNEL_GM_C2:
        return c * c / (c * c + e2) ** 2
    return np.ones_like(e2)


class VoxelMap:
    """mola::HashedVoxelPointCloud semantics (lidar3d-default.yaml:228-242)."""

    def __init__(self, voxel_size=1.0, max_points_per_voxel=20, trunc=False):
        self.inv = np.float32(1.0) / np.float32(voxel_size)
        self.cap = max_points_per_voxel
        self.trunc = trunc
        self.vox: dict[tuple, list] = {}
        self.n_offered = 0

    def key(self, p):
        s = np.asarray(p, dtype=np.float32) * self.inv
        k = np.trunc(s) if self.trunc else np.floor(s)
        return tuple(int(v) for v in k)

    def insert(self, xyz):
        xyz = np.asarray(xyz, dtype=np.float32)
        for i, p in enumerate(xyz):
            if not np.all(np.isfinite(p)):
                continue
            lst = self.vox.setdefault(self.key(p), [])
            if self.cap and len(lst) >= self.cap:
                continue
            lst.append((self.n_offered + i, p.copy()))
        self.n_offered += len(xyz)
        return self

    @property
    def num_points(self):
        return sum(len(v) for v in self.vox.values())

    def nn_single(self, q):
        """27-block search, x outer / y middle / z inner, first strict minimum."""
        q = np.asarray(q, dtype=np.float32)
        c = self.key(q)
        cand_idx, cand_pts = [], []
        for ix in (c[0] - 1, c[0], c[0] + 1):
            for iy in (c[1] - 1, c[1], c[1] + 1):
                for iz in (c[2] - 1, c[2], c[2] + 1):
                    for (i, p) in self.vox.get((ix, iy, iz), ()):
                        cand_idx.append(i)
                        cand_pts.append(p)
        if not cand_idx:
            return None
        P = np.stack(cand_pts).astype(np.float32)
        d = P - q[None, :]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]  # float32, same order as the spec
        j = int(np.argmin(d2))  # argmin returns the first minimum
        return cand_idx[j], P[j], np.float32(d2[j]), len(cand_idx)


def transform_points_f32(T, l):
    """p' = (float)(R*l + t) with double pose (Matcher_Points_Base::transform_local_to_global)."""
    T = np.asarray(T, dtype=np.float64)
    l64 = np.asarray(l, dtype=np.float32).astype(np.float64)
    R, t = T[:3, :3], T[:3, 3]
    out = np.empty_like(l64)
    for r in range(3):
        out[:, r] = ((R[r, 0] * l64[:, 0] + R[r, 1] * l64[:, 1]) + R[r, 2] * l64[:, 2]) + t[r]
    return out.astype(np.float32)


def match_points(m: VoxelMap, local_xyz, T, threshold, threshold_angular_deg=0.0):
    """Matcher_Points_DistanceThreshold, pairingsPerPoint=1 (lidar3d-default.yaml:195-204)."""
    l = np.asarray(local_xyz, dtype=np.float32)
    g = transform_points_f32(T, l)
    thr2 = np.float32(threshold * threshold)
    ang2 = np.float32(np.deg2rad(threshold_angular_deg) ** 2)
    li, gi, gp, dd = [], [], [], []
    ncand = 0
    for i, p in enumerate(g):
        r = m.nn_single(p)
        if r is None:
            continue
        idx, q, d2, nc = r
        ncand += nc
        n2 = np.float32(np.float32(p[0] * p[0] + p[1] * p[1]) + p[2] * p[2])
        if d2 < np.float32(thr2 + ang2 * n2):
            li.append(i); gi.append(idx); gp.append(q); dd.append(d2)
    return dict(local_idx=np.array(li, np.uint32), global_idx=np.array(gi, np.uint32),
                global_xyz=np.array(gp, np.float32).reshape(-1, 3), d2=np.array(dd, np.float32),
                potential_pairings=len(l), n_candidates=ncand)


def dDexpe_de(T):
    """12x6 Jacobian of vec([d1 d2 d3 t]) of T*exp(eps) wrt eps=[v;w] at 0 (MRPT jacob_dDexpe_de)."""
    R = T[:3, :3]
    J = np.zeros((12, 6))
    for j in range(3):
        ej = np.zeros(3); ej[j] = 1.0
        J[3 * j:3 * j + 3, 3:6] = -R @ hat(ej)  # d(R e_j)/dw
    J[9:12, 0:3] = R
    return J


def accumulate(T, pt2pt=None, pt2pl=None, kernel=KERNEL_GM_C4, c=1.0, w_pt2pt=1.0, w_pt2pl=1.0):
    """H = sum w J^T J, g = sum w J^T e  (optimal_tf_gauss_newton)."""
    T = np.asarray(T, dtype=np.float64)
    R, t = T[:3, :3], T[:3, 3]
    H, g, cost = np.zeros((6, 6)), np.zeros(6), 0.0
    D = dDexpe_de(T)
    if pt2pt is not None:
        L = np.asarray(pt2pt[0], np.float32).astype(np.float64).reshape(-1, 3)
        Q = np.asarray(pt2pt[1], np.float32).astype(np.float64).reshape(-1, 3)
        for l, q in zip(L, Q):
            e = R @ l + t - q
            J1 = np.hstack([l[0] * np.eye(3), l[1] * np.eye(3), l[2] * np.eye(3), np.eye(3)])  # 3x12
            J = J1 @ D
            e2 = float(e @ e)
            w = w_pt2pt * float(robust_weight(kernel, c, e2))
            H += w * J.T @ J
            g += w * J.T @ e
            cost += w * e2
    if pt2pl is not None:
        L = np.asarray(pt2pl[0], np.float32).astype(np.float64).reshape(-1, 3)
        Cc = np.asarray(pt2pl[1], np.float32).astype(np.float64).reshape(-1, 3)
        N = np.asarray(pt2pl[2], np.float32).astype(np.float64).reshape(-1, 3)
        for l, cc, n in zip(L, Cc, N):
            e = float(n @ (R @ l + t - cc))
            J1 = np.hstack([l[0] * np.eye(3), l[1] * np.eye(3), l[2] * np.eye(3), np.eye(3)])
            J = (n[None, :] @ J1 @ D).reshape(6)
            w = w_pt2pl * float(robust_weight(kernel, c, e * e))
            H += w * np.outer(J, J)
            g += w * J * e
            cost += w * e * e
    return H, g, cost


def prior_term(prior, T):
    """e_p = log(T_p^-1 T), Jp by central differences on T*exp(eps)."""
    Tp, Lam = np.asarray(prior[0], np.float64), np.asarray(prior[1], np.float64).reshape(6, 6)
    if Tp.size == 12:
        Tp = T44(Tp)
    D = np.linalg.inv(Tp) @ T
    e0 = se3_log(D)
    h = 1e-6
    Jp = np.zeros((6, 6))
    for j in range(6):
        d = np.zeros(6); d[j] = h
        Jp[:, j] = (se3_log(D @ se3_exp(d)) - se3_log(D @ se3_exp(-d))) / (2 * h)
    return Jp.T @ Lam @ Jp, Jp.T @ Lam @ e0


def gn_solve(T, pt2pt=None, pt2pl=None, inner=2, kernel=KERNEL_GM_C4, c=1.0, prior=None, min_delta=1e-7,
             max_cost=0.0):
    T = np.asarray(T, dtype=np.float64).copy()
    steps = []
    for _ in range(inner):
        H, g, cost = accumulate(T, pt2pt, pt2pl, kernel, c)
        if prior is not None:
            Hp, gp = prior_term(prior, T)
            H, g = H + Hp, g + gp
        if np.sqrt(cost) <= max_cost:
            steps.append(dict(H=H, g=g, cost=cost, delta=np.zeros(6)))
            break
        delta = -np.linalg.solve(H, g)
        T = T @ se3_exp(delta)
        steps.append(dict(H=H, g=g, cost=cost, delta=delta))
        if np.linalg.norm(delta) < min_delta:
            break
    return T, steps


def covariance(T, pt2pt, findif_xyz=1e-7, findif_ang=1e-7):
    """mp2p_icp::covariance: numeric Jacobian wrt (x,y,z,yaw,pitch,roll)."""
    L = np.asarray(pt2pt[0], np.float32).astype(np.float64).reshape(-1, 3)
    if len(L) == 0:
        return np.eye(6) * 1e6
    T = np.asarray(T, np.float64)
    R = T[:3, :3]
    pitch = np.arctan2(-R[2, 0], np.hypot(R[0, 0], R[1, 0]))
    yaw = np.arctan2(R[1, 0], R[0, 0])
    roll = np.arctan2(R[2, 1], R[2, 2])
    x0 = np.array([T[0, 3], T[1, 3], T[2, 3], yaw, pitch, roll])

    def resid(x):
        Tx = pose_from_ypr(x)
        return (L @ Tx[:3, :3].T + Tx[:3, 3]).reshape(-1)

    A = np.zeros((3 * len(L), 6))
    for j in range(6):
        h = findif_xyz if j < 3 else findif_ang
        d = np.zeros(6); d[j] = h
        A[:, j] = (resid(x0 + d) - resid(x0 - d)) / (2 * h)
    return np.linalg.inv(A.T @ A)


def icp_align(m: VoxelMap, local_xyz, T_guess, thresholds, kernel_params, max_iterations, inner=2,
              kernel=KERNEL_GM_C4, min_step_trans=1e-4, min_step_rot=5e-5, disable_stall=False, prior=None):
    """ICP::align outer loop (call site LidarOdometry.cpp:961-962)."""
    l = np.asarray(local_xyz, dtype=np.float32)
    T = np.asarray(T_guess, dtype=np.float64).copy()
    if T.size == 12:
        T = T44(T)
    Tprev = T.copy()
    reason, pairs, trace = "Undefined", None, []
    it = 0
    while it < max_iterations:
        pairs = match_points(m, l, T, thresholds[it])
        if len(pairs["local_idx"]) == 0:
            reason = "NoPairings"
            break
        T, _ = gn_solve(T, (l[pairs["local_idx"]], pairs["global_xyz"]), None, inner, kernel, kernel_params[it],
                        prior)
        d = se3_log(np.linalg.inv(Tprev) @ T)
        trace.append(dict(T=T.copy(), n_pairs=len(pairs["local_idx"])))
        if (not disable_stall) and np.linalg.norm(d[:3]) < min_step_trans and np.linalg.norm(d[3:]) < min_step_rot:
            reason = "Stalled"
            break
        Tprev = T.copy()
        it += 1
    if it >= max_iterations:
        reason = "MaxIterations"
    npairs = 0 if pairs is None else len(pairs["local_idx"])
    quality = npairs / len(l) if len(l) and npairs else 0.0
    return dict(T=T, n_iterations=it, termination_reason=reason, quality=quality, pairs=pairs, trace=trace)
