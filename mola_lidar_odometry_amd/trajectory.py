"""Trajectory I/O and evaluation for the stand-alone odometry driver.

Formats the reference's evaluation scripts exchange (eval/cli_kitti.sh:41-50, relative to /root/reference; test fixtures
test/kitti_00_fragment_gt.tum, test/rslidar_fragment_gt.tum): TUM text trajectories "t x y z qx qy qz qw", KITTI
velodyne .bin scans (float32 x,y,z,intensity), and the two error measures it quotes: absolute trajectory error and the
KITTI odometry benchmark's relative translation / rotation errors (the metric `kitti-metrics-eval` [U] reports).
Pure numpy; host-side bookkeeping only.
"""
from __future__ import annotations

import numpy as np


# ------------------------------------------------------------------------------------------------ poses
def quat_to_rot(q) -> np.ndarray:
    """(qx,qy,qz,qw) -> 3x3"""
    x, y, z, w = [float(v) for v in q]
    n = np.sqrt(x * x + y * y + z * z + w * w)
    x, y, z, w = x / n, y / n, z / n, w / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def rot_to_quat(R) -> np.ndarray:
    """3x3 -> (qx,qy,qz,qw), qw >= 0"""
    R = np.asarray(R, np.float64)
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s, (R[2, 1] - R[1, 2]) / s]
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s, (R[0, 2] - R[2, 0]) / s]
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s, (R[1, 0] - R[0, 1]) / s]
    q = np.asarray(q)
    return -q if q[3] < 0 else q


def to44(T12) -> np.ndarray:
    return np.vstack([np.asarray(T12, np.float64).reshape(3, 4), [0, 0, 0, 1]])


# ------------------------------------------------------------------------------------------------ TUM files
def read_tum(path):
    """-> (stamps [N], poses [N,4,4])"""
    rows = [l.split() for l in open(path) if l.strip() and not l.startswith("#")]
    a = np.asarray(rows, dtype=np.float64).reshape(-1, 8)
    T = np.tile(np.eye(4), (len(a), 1, 1))
    for i, r in enumerate(a):
        T[i, :3, :3] = quat_to_rot(r[4:8])
        T[i, :3, 3] = r[1:4]
    return a[:, 0].copy(), T


def write_tum(path, stamps, poses):
    with open(path, "w") as f:
        for t, T in zip(stamps, poses):
            T = np.asarray(T, np.float64)
            T = T.reshape(3, 4) if T.size == 12 else T[:3]
            q = rot_to_quat(T[:, :3])
            f.write("%.9f %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n" % (t, T[0, 3], T[1, 3], T[2, 3], q[0], q[1], q[2], q[3]))


# ------------------------------------------------------------------------------------------------ KITTI scans
def read_kitti_bin(path):
    """KITTI odometry velodyne scan -> (xyz [N,3] fp32, intensity [N] fp32)."""
    a = np.fromfile(path, dtype=np.float32).reshape(-1, 4)
    return np.ascontiguousarray(a[:, :3]), np.ascontiguousarray(a[:, 3])


def kitti_azimuth_timestamps(xyz, sweep_time=0.1):
    """Per-point time stamps for a KITTI scan, which stores none: the HDL-64E spins clockwise seen from above, one
    revolution per scan, so the firing time follows the azimuth; centred on zero (TimestampAdjustMethod::MiddleIsZero).
    KITTI scans are motion compensated already -- use with skip_deskew unless the raw (unsynced) data is fed."""
    az = np.arctan2(xyz[:, 1], xyz[:, 0])
    return (-(az / (2 * np.pi)) * sweep_time).astype(np.float32)


# ------------------------------------------------------------------------------------------------ errors
def associate(stamps_a, stamps_b, max_dt=0.02):
    """Nearest-time association -> index pairs (ia, ib)."""
    ib = np.searchsorted(stamps_b, stamps_a)
    ib = np.clip(ib, 1, len(stamps_b) - 1)
    left = np.abs(stamps_b[ib - 1] - stamps_a) < np.abs(stamps_b[ib] - stamps_a)
    ib = np.where(left, ib - 1, ib)
    ok = np.abs(stamps_b[ib] - stamps_a) <= max_dt
    return np.nonzero(ok)[0], ib[ok]


def umeyama_se3(src, dst):
    """Rigid (R,t) minimising sum |R src + t - dst|^2."""
    mu_s, mu_d = src.mean(0), dst.mean(0)
    H = (dst - mu_d).T @ (src - mu_s)
    U, _, Vt = np.linalg.svd(H)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
    R = U @ D @ Vt
    return R, mu_d - R @ mu_s


def ate_rmse(est, gt, align="origin"):
    """Absolute trajectory error [m] (RMSE of position differences).  align: 'origin' expresses both trajectories
    relative to their first pose (what test/test_lidar_odometry_rawlog.cpp:95-104 does), 'se3' fits a rigid
    transformation first (evo's -a), 'none' compares as given."""
    est, gt = np.asarray(est, np.float64), np.asarray(gt, np.float64)
    if align == "origin":
        est = np.linalg.inv(est[0])[None] @ est
        gt = np.linalg.inv(gt[0])[None] @ gt
    pe, pg = est[:, :3, 3], gt[:, :3, 3]
    if align == "se3":
        R, t = umeyama_se3(pe, pg)
        pe = pe @ R.T + t
    return float(np.sqrt(np.mean(np.sum((pe - pg) ** 2, axis=1))))


def kitti_relative_errors(est, gt, lengths=(100, 200, 300, 400, 500, 600, 700, 800), step=10):
    """KITTI odometry benchmark errors: for every `step`-th start frame and every path length, the pose error of the
    sub-trajectory of that length: translation [% of length] and rotation [deg/m], averaged.  Returns
    (t_err_percent, r_err_deg_per_m, n_segments); NaN when the trajectory is shorter than the smallest length."""
    est, gt = np.asarray(est, np.float64), np.asarray(gt, np.float64)
    step_d = np.linalg.norm(np.diff(gt[:, :3, 3], axis=0), axis=1)
    dist = np.concatenate([[0.0], np.cumsum(step_d)])
    te, re = [], []
    for first in range(0, len(gt), step):
        for L in lengths:
            last = int(np.searchsorted(dist, dist[first] + L))
            if last >= len(gt):
                continue
            d_gt = np.linalg.inv(gt[first]) @ gt[last]
            d_est = np.linalg.inv(est[first]) @ est[last]
            E = np.linalg.inv(d_est) @ d_gt
            te.append(np.linalg.norm(E[:3, 3]) / L)
            c = np.clip(0.5 * (np.trace(E[:3, :3]) - 1.0), -1.0, 1.0)
            re.append(np.arccos(c) / L)
    if not te:
        return float("nan"), float("nan"), 0
    return 100.0 * float(np.mean(te)), float(np.degrees(np.mean(re))), len(te)
