"""Deterministic synthetic workloads for the ICP hot path.

There is no network and the reference bundles no point clouds, so every bench and
parity input is generated here: a "street canyon" scene (ground plane, two facades at y=+-12 m,
end walls, random boxes), a local map sampled on its surfaces and pushed through the voxel-cap
rule of mola::HashedVoxelPointCloud (lidar3d-default.yaml:233-235), and an HDL-64-like scan
ray-cast from a known pose.  Pure numpy; PCG64 with fixed seeds, so inputs are identical here and
on the GPU box.

C2 (BASELINE.json configs[1]): workload_c2() -> N = 120 000 scan points vs M = 1 000 000 map
points, voxel 1.0 m, cap 20, 20 ICP iterations with the sigma=2.0 schedule of
lidar3d-default.yaml:190,198.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

FACADE_Y = 12.0
FACADE_H = 15.0
SENSOR_H = 1.73


@dataclass
class Scene:
    half_extent: float  # ground is [-h, h]^2
    end_wall_x: float   # end walls at x = +-end_wall_x (|y| <= FACADE_Y)
    boxes: np.ndarray   # [B, 6] = xmin, ymin, zmin, xmax, ymax, zmax


def make_scene(seed: int = 12345, half_extent: float = 120.0, n_boxes: int = 40) -> Scene:
    rng = np.random.Generator(np.random.PCG64(seed))
    end_wall_x = half_extent - 10.0
    boxes = []
    while len(boxes) < n_boxes:
        cx = rng.uniform(-end_wall_x + 6.0, end_wall_x - 6.0)
        cy = rng.uniform(-FACADE_Y + 2.0, FACADE_Y - 2.0)
        if rng.uniform() < 0.6:  # "car"
            sx, sy, sz = rng.uniform(3.5, 4.8), rng.uniform(1.6, 2.0), rng.uniform(1.4, 1.9)
        else:  # kiosk / small building
            sx, sy, sz = rng.uniform(2.0, 8.0), rng.uniform(1.5, 3.0), rng.uniform(2.5, 6.0)
        if np.hypot(cx, cy) < 8.0:  # keep the sensor's surroundings free
            continue
        boxes.append([cx - sx / 2, cy - sy / 2, 0.0, cx + sx / 2, cy + sy / 2, sz])
    return Scene(half_extent, end_wall_x, np.asarray(boxes, dtype=np.float64))


def _sample_surfaces(scene: Scene, n: int, rng: np.random.Generator, noise: float) -> np.ndarray:
    h, ex = scene.half_extent, scene.end_wall_x
    # surface list: (area, sampler)
    areas, samplers = [], []

    def add(area, fn):
        areas.append(area)
        samplers.append(fn)

    add((2 * h) ** 2, lambda k: np.stack([rng.uniform(-h, h, k), rng.uniform(-h, h, k), np.zeros(k)], 1))
    for s in (-1.0, 1.0):
        add(2 * h * FACADE_H,
            lambda k, s=s: np.stack([rng.uniform(-h, h, k), np.full(k, s * FACADE_Y), rng.uniform(0, FACADE_H, k)], 1))
        add(2 * FACADE_Y * FACADE_H,
            lambda k, s=s: np.stack([np.full(k, s * ex), rng.uniform(-FACADE_Y, FACADE_Y, k),
                                     rng.uniform(0, FACADE_H, k)], 1))
    for b in scene.boxes:
        sx, sy, sz = b[3] - b[0], b[4] - b[1], b[5] - b[2]
        # 4 sides + top, sampled as one surface by face areas
        def box_fn(k, b=b, sx=sx, sy=sy, sz=sz):
            fa = np.array([sx * sz, sx * sz, sy * sz, sy * sz, sx * sy])
            f = rng.choice(5, size=k, p=fa / fa.sum())
            u, v = rng.uniform(0, 1, k), rng.uniform(0, 1, k)
            p = np.zeros((k, 3))
            m = f == 0; p[m] = np.stack([b[0] + u[m] * sx, np.full(m.sum(), b[1]), b[2] + v[m] * sz], 1)
            m = f == 1; p[m] = np.stack([b[0] + u[m] * sx, np.full(m.sum(), b[4]), b[2] + v[m] * sz], 1)
            m = f == 2; p[m] = np.stack([np.full(m.sum(), b[0]), b[1] + u[m] * sy, b[2] + v[m] * sz], 1)
            m = f == 3; p[m] = np.stack([np.full(m.sum(), b[3]), b[1] + u[m] * sy, b[2] + v[m] * sz], 1)
            m = f == 4; p[m] = np.stack([b[0] + u[m] * sx, b[1] + v[m] * sy, np.full(m.sum(), b[5])], 1)
            return p
        add(2 * (sx + sy) * sz + sx * sy, box_fn)
    areas = np.asarray(areas)
    # vertical structure is what a lidar map is dense in: weight non-ground surfaces x4
    wts = areas.copy()
    wts[1:] *= 4.0
    counts = rng.multinomial(n, wts / wts.sum())
    pts = np.concatenate([fn(int(k)) for fn, k in zip(samplers, counts) if k > 0], 0)
    pts += rng.normal(0.0, noise, pts.shape)
    return pts[rng.permutation(len(pts))]


def voxel_cap_filter(xyz: np.ndarray, voxel_size: float, cap: int) -> np.ndarray:
    """Boolean mask of the points insertPoint() would keep (first `cap` per voxel, in order).
    Data generation only -- NOT the product map build and NOT the oracle."""
    k = np.floor(xyz.astype(np.float32) * (np.float32(1.0) / np.float32(voxel_size))).astype(np.int64)
    key = ((k[:, 0] + (1 << 20)) << 42) | ((k[:, 1] + (1 << 20)) << 21) | (k[:, 2] + (1 << 20))
    order = np.argsort(key, kind="stable")
    ks = key[order]
    head = np.ones(len(ks), dtype=bool)
    head[1:] = ks[1:] != ks[:-1]
    start = np.maximum.accumulate(np.where(head, np.arange(len(ks)), 0))
    rank = np.arange(len(ks)) - start
    keep = np.zeros(len(ks), dtype=bool)
    keep[order] = rank < cap if cap else True
    return keep


def make_map(scene: Scene, n_points: int, seed: int = 12345, voxel_size: float = 1.0, cap: int = 20,
             noise: float = 0.02) -> np.ndarray:
    """Exactly n_points fp32 map points, all of which survive the voxel-cap rule in this order."""
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    kept = np.zeros((0, 3), dtype=np.float32)
    cand = np.zeros((0, 3), dtype=np.float32)
    factor = 1.6
    for _ in range(8):
        cand = np.concatenate([cand, _sample_surfaces(scene, int(n_points * factor) - len(cand), rng, noise)
                               .astype(np.float32)], 0)
        keep = voxel_cap_filter(cand, voxel_size, cap)
        if keep.sum() >= n_points:
            kept = cand[keep][:n_points]
            break
        factor *= 1.5
    if len(kept) != n_points:
        raise RuntimeError("scene too small for the requested number of map points under the voxel cap")
    return np.ascontiguousarray(kept, dtype=np.float32)


def pose_from_ypr(p) -> np.ndarray:
    """TPose3D (x,y,z,yaw,pitch,roll) -> row-major 3x4 [R|t] flattened (12 doubles), R=Rz*Ry*Rx."""
    x, y, z, yaw, pitch, roll = [float(v) for v in p]
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    return np.array([cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr, x,
                     sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr, y,
                     -sp, cp * sr, cp * cr, z], dtype=np.float64)


def make_scan(scene: Scene, pose_ypr, rings: int = 64, azimuths: int = 1875, seed: int = 54321,
              max_range: float = 120.0, range_noise: float = 0.02) -> np.ndarray:
    """Ray-cast an HDL-64-like sweep (ring-major order) from pose_ypr; returns LOCAL-frame fp32 points."""
    rng = np.random.Generator(np.random.PCG64(seed))
    T = pose_from_ypr(pose_ypr).reshape(3, 4)
    R, o = T[:, :3], T[:, 3]
    el = np.deg2rad(np.linspace(2.0, -24.8, rings))
    az = np.linspace(-np.pi, np.pi, azimuths, endpoint=False)
    E, A = np.meshgrid(el, az, indexing="ij")
    d_local = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    d = d_local @ R.T
    tmin = np.full(len(d), np.inf)
    h, ex = scene.half_extent, scene.end_wall_x

    def plane(axis, value, bounds):
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (value - o[axis]) / d[:, axis]
        p = o[None, :] + t[:, None] * d
        ok = (t > 1e-6) & np.isfinite(t)
        for ax, lo, hi in bounds:
            ok &= (p[:, ax] >= lo) & (p[:, ax] <= hi)
        np.minimum(tmin, np.where(ok, t, np.inf), out=tmin)

    plane(2, 0.0, [(0, -h, h), (1, -h, h)])
    for s in (-1.0, 1.0):
        plane(1, s * FACADE_Y, [(0, -h, h), (2, 0.0, FACADE_H)])
        plane(0, s * ex, [(1, -FACADE_Y, FACADE_Y), (2, 0.0, FACADE_H)])
    for b in scene.boxes:  # slab test
        with np.errstate(divide="ignore", invalid="ignore"):
            t0 = (b[None, :3] - o[None, :]) / d
            t1 = (b[None, 3:] - o[None, :]) / d
        tn = np.nanmax(np.minimum(t0, t1), axis=1)
        tf = np.nanmin(np.maximum(t0, t1), axis=1)
        ok = (tf >= tn) & (tn > 1e-6)
        np.minimum(tmin, np.where(ok, tn, np.inf), out=tmin)
    hit = tmin < max_range
    rngs = tmin[hit] + rng.normal(0.0, range_noise, int(hit.sum()))
    return np.ascontiguousarray(d_local[hit] * rngs[:, None], dtype=np.float32)


def _so3_exp(w: np.ndarray) -> np.ndarray:
    """Rodrigues for a batch of rotation vectors [N,3] -> [N,3,3]."""
    th = np.linalg.norm(w, axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        a = np.where(th < 1e-8, 1.0 - th * th / 6.0, np.sin(th) / th)
        b = np.where(th < 1e-8, 0.5 - th * th / 24.0, (1.0 - np.cos(th)) / (th * th))
    W = np.zeros((len(w), 3, 3))
    W[:, 0, 1], W[:, 0, 2], W[:, 1, 0], W[:, 1, 2], W[:, 2, 0], W[:, 2, 1] = -w[:, 2], w[:, 1], w[:, 2], -w[:, 0], -w[:, 1], w[:, 0]
    return np.eye(3)[None] + a[:, None, None] * W + b[:, None, None] * (W @ W)


def _raycast(scene: Scene, o: np.ndarray, d: np.ndarray) -> np.ndarray:
    """Distance along each ray (origins o[N,3], unit directions d[N,3]) to the first scene surface (inf = none)."""
    tmin = np.full(len(d), np.inf)
    h, ex = scene.half_extent, scene.end_wall_x

    def plane(axis, value, bounds):
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (value - o[:, axis]) / d[:, axis]
        p = o + t[:, None] * d
        ok = (t > 1e-6) & np.isfinite(t)
        for ax, lo, hi in bounds:
            ok &= (p[:, ax] >= lo) & (p[:, ax] <= hi)
        np.minimum(tmin, np.where(ok, t, np.inf), out=tmin)

    plane(2, 0.0, [(0, -h, h), (1, -h, h)])
    for s in (-1.0, 1.0):
        plane(1, s * FACADE_Y, [(0, -h, h), (2, 0.0, FACADE_H)])
        plane(0, s * ex, [(1, -FACADE_Y, FACADE_Y), (2, 0.0, FACADE_H)])
    for b in scene.boxes:  # slab test
        with np.errstate(divide="ignore", invalid="ignore"):
            t0 = (b[None, :3] - o) / d
            t1 = (b[None, 3:] - o) / d
        tn = np.nanmax(np.minimum(t0, t1), axis=1)
        tf = np.nanmin(np.maximum(t0, t1), axis=1)
        ok = (tf >= tn) & (tn > 1e-6)
        np.minimum(tmin, np.where(ok, tn, np.inf), out=tmin)
    return tmin


def make_sweep(scene: Scene, T_ref, twist, sweep_time: float = 0.1, rings: int = 32, azimuths: int = 600,
               seed: int = 1, max_range: float = 120.0, range_noise: float = 0.02):
    """One sweep of a spinning LiDAR that moves with a constant twist (vx,vy,vz,wx,wy,wz, vehicle frame) while it
    scans: azimuth column j fires at t_j = (j/azimuths - 0.5)*sweep_time relative to the scan's reference time, from
    the pose T_ref (+) (Exp_SO3(w t_j), v t_j) -- the motion model FilterDeskew undoes (lidar3d-default.yaml:328-350).
    Returns (xyz [N,3] fp32 in the sensor frame AT FIRING TIME, i.e. skewed; t [N] fp32), ring-major order."""
    rng = np.random.Generator(np.random.PCG64(seed))
    T = np.asarray(T_ref, dtype=np.float64).reshape(3, 4)
    tw = np.asarray(twist, dtype=np.float64)
    el = np.deg2rad(np.linspace(2.0, -24.8, rings))
    az = np.linspace(-np.pi, np.pi, azimuths, endpoint=False)
    tcol = (np.arange(azimuths) / azimuths - 0.5) * sweep_time
    E, A = np.meshgrid(el, az, indexing="ij")
    tt = np.broadcast_to(tcol[None, :], E.shape).reshape(-1)
    d_local = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    Rt = _so3_exp(tw[None, 3:] * tt[:, None])                      # rotation of the sensor at firing time, in T_ref
    pt = tw[None, :3] * tt[:, None]
    R = T[None, :, :3] @ Rt
    o = (T[:, :3] @ pt.T).T + T[:, 3]
    d = np.einsum("nij,nj->ni", R, d_local)
    tmin = _raycast(scene, o, d)
    hit = tmin < max_range
    rngs = tmin[hit] + rng.normal(0.0, range_noise, int(hit.sum()))
    return (np.ascontiguousarray(d_local[hit] * rngs[:, None], dtype=np.float32),
            np.ascontiguousarray(tt[hit], dtype=np.float32))


def drive_plan(n_scans: int = 12, dt: float = 0.1, speed: float = 8.0, yaw_rate: float = 0.12, seed: int = 4242,
               ramp_scans: int = 6):
    """Poses, twists, stamps and sweep seeds of make_drive (cheap: a recurrence); the sweeps themselves are independent
    given these, so a caller may cast them in worker processes (drive_sweep)."""
    T = pose_from_ypr([-60.0, 0.5, SENSOR_H, 0.0, 0.0, 0.0]).reshape(3, 4)
    poses, twists, stamps, seeds = [], [], [], []
    for k in range(n_scans):
        gain = min(1.0, k / float(ramp_scans)) if ramp_scans > 0 else 1.0
        tw = np.array([gain * speed * (1.0 + 0.05 * np.sin(0.7 * k)), 0.02 * np.cos(0.5 * k), 0.0, 0.0, 0.0,
                       gain * yaw_rate * np.sin(0.9 * k)])
        poses.append(T.reshape(12).copy())
        twists.append(tw)
        stamps.append(1000.0 + k * dt)
        seeds.append(seed + 10 * k)
        Rn = _so3_exp((tw[3:] * dt)[None])[0]
        T = np.concatenate([T[:, :3] @ Rn, (T[:, :3] @ (tw[:3] * dt) + T[:, 3])[:, None]], axis=1)
    return dict(poses=np.asarray(poses), twists=np.asarray(twists), stamps=np.asarray(stamps), seeds=seeds, dt=dt)


def drive_sweep(args):
    """One sweep of a drive_plan: args = (scene seed, half_extent, n_boxes, pose, twist, dt, rings, azimuths, sweep seed)."""
    scene_seed, half_extent, n_boxes, pose, tw, dt, rings, azimuths, sweep_seed = args
    return make_sweep(make_scene(scene_seed, half_extent, n_boxes), pose, tw, dt, rings, azimuths, sweep_seed)


def make_drive(n_scans: int = 12, dt: float = 0.1, speed: float = 8.0, yaw_rate: float = 0.12, rings: int = 32,
               azimuths: int = 600, seed: int = 4242, half_extent: float = 120.0, n_boxes: int = 160,
               ramp_scans: int = 6, pool=None):
    """A synthetic drive along the street canyon: ground-truth poses (12 doubles each), per-scan twists and skewed
    sweeps with per-point time stamps.  The vehicle pulls away from rest over `ramp_scans` scans and then moves with
    a slowly varying twist; the pose at scan k+1 is the pose at scan k composed with (Exp_SO3(w dt), v dt), so a
    constant-velocity model is exact up to the variation.  The canyon is cluttered (n_boxes) because two facades and
    a ground plane alone do not constrain the motion along the street.  `pool`: a multiprocessing pool to cast the
    sweeps in (same result)."""
    scene = make_scene(seed, half_extent, n_boxes)
    plan = drive_plan(n_scans, dt, speed, yaw_rate, seed, ramp_scans)
    if pool is not None:
        scans = pool.map(drive_sweep, [(seed, half_extent, n_boxes, plan["poses"][k], plan["twists"][k], dt, rings, azimuths,
                                        plan["seeds"][k]) for k in range(n_scans)])
    else:
        scans = [make_sweep(scene, plan["poses"][k], plan["twists"][k], dt, rings, azimuths, plan["seeds"][k])
                 for k in range(n_scans)]
    return dict(scene=scene, poses=plan["poses"], twists=plan["twists"], scans=scans, stamps=plan["stamps"])


def write_kitti_sequence(root: str, drive, seq: str = "00") -> str:
    """The drive as a KITTI odometry sequence folder (velodyne/%06d.bin rows of float32 x,y,z,intensity + times.txt), what
    molahip-lo-cli and eval/cli_kitti.sh's --input-kitti-seq read.  Returns the sequence directory."""
    import os
    d = os.path.join(root, "sequences", seq)
    os.makedirs(os.path.join(d, "velodyne"), exist_ok=True)
    for k, (xyz, _) in enumerate(drive["scans"]):
        np.concatenate([xyz, np.zeros((len(xyz), 1), np.float32)], 1).astype(np.float32).tofile(
            os.path.join(d, "velodyne", "%06d.bin" % k))
    np.savetxt(os.path.join(d, "times.txt"), drive["stamps"] - drive["stamps"][0], fmt="%.6e")
    return d


def ndt_cloud(seed=0):
    """Ground plane + wall + blob with 1 cm noise: planar and non-planar voxels for the NDT map / point-to-plane tests."""
    rng = np.random.default_rng(seed)
    ground = np.stack([rng.uniform(-10, 10, 20000), rng.uniform(-10, 10, 20000), rng.normal(0.3, 0.01, 20000)], 1)
    wall = np.stack([rng.uniform(-10, 10, 12000), rng.normal(5.4, 0.01, 12000), rng.uniform(0.5, 4, 12000)], 1)
    blob = rng.normal([3.5, -3.5, 1.5], 0.25, (4000, 3))
    return np.concatenate([ground, wall, blob]).astype(np.float32)


def threshold_schedule(sigma: float, n_iters: int):
    """Matcher threshold and robust-kernel parameter as functions of ICP_ITERATION
    (lidar3d-default.yaml:198 and :190)."""
    k = np.arange(n_iters, dtype=np.float64)
    base = np.maximum(sigma, 2.0 * sigma - (2.0 * sigma - 0.5 * sigma) * k / 30.0)
    return 2.0 * base, 0.5 * base


@dataclass
class Workload:
    name: str
    map_xyz: np.ndarray      # [M,3] fp32
    scan_xyz: np.ndarray     # [N,3] fp32, local frame
    pose_gt_ypr: np.ndarray  # TPose3D
    guess_ypr: np.ndarray
    T_gt: np.ndarray         # 12 doubles
    T_guess: np.ndarray
    voxel_size: float
    cap: int
    n_iters: int
    threshold: np.ndarray
    kernel_param: np.ndarray
    sigma: float


GUESS_PERTURBATION = np.array([0.5, 0.1, 0.02, np.deg2rad(1.0), np.deg2rad(0.2), np.deg2rad(0.2)])
POSE_GT = np.array([2.5, -1.2, SENSOR_H, np.deg2rad(8.0), np.deg2rad(0.5), np.deg2rad(-0.3)])


def make_workload(name: str, n_map: int, rings: int, azimuths: int, half_extent: float, n_boxes: int,
                  n_iters: int = 20, sigma: float = 2.0, voxel_size: float = 1.0, cap: int = 20,
                  variant: int = 0) -> Workload:
    """variant 0 is the canonical workload (the golden fixtures and workload_stats.json refer to it); variant j > 0 is
    an independent draw of the same generator -- another scene (boxes), another map sample, another sensor pose and
    scan noise -- so that a batch of S scans works on S different maps, as S sequences would (eval/cli_kitti.sh:23-36)."""
    scene = make_scene(12345 + 1009 * variant, half_extent, n_boxes)
    m = make_map(scene, n_map, 12345 + 1009 * variant, voxel_size, cap)
    pose_gt = POSE_GT.copy()
    if variant:
        rng = np.random.Generator(np.random.PCG64(777 + variant))
        pose_gt[:2] += rng.uniform(-1.5, 1.5, 2) * np.array([4.0, 1.0])
        pose_gt[3] += rng.uniform(-0.3, 0.3)
    s = make_scan(scene, pose_gt, rings, azimuths, 54321 + 31 * variant)
    guess = pose_gt + GUESS_PERTURBATION
    thr, kp = threshold_schedule(sigma, n_iters)
    return Workload(name if not variant else f"{name}#{variant}", m, s, pose_gt, guess, pose_from_ypr(pose_gt),
                    pose_from_ypr(guess), voxel_size, cap, n_iters, thr, kp, sigma)


def workload_c2(variant: int = 0) -> Workload:
    """BASELINE.json configs[1]: ~120k-pt scan vs 1M-pt map, 20 iterations."""
    return make_workload("C2_120k_vs_1M", 1_000_000, 64, 1875, 120.0, 40, variant=variant)


def workload_small(variant: int = 0) -> Workload:
    """Reduced copy of C2 (2k-pt scan vs 20k-pt map): the committed golden fixture's generator."""
    return make_workload("small_2k_vs_20k", 20_000, 16, 125, 25.0, 6, variant=variant)


def workload_creal(variant: int = 0) -> Workload:
    """What lidar3d-default.yaml actually feeds align(): a few-thousand-point decimated scan
    (yaml:285-319) against the same 1M-pt map."""
    return make_workload("Creal_6k_vs_1M", 1_000_000, 32, 192, 120.0, 40, variant=variant)


def workload_by_name(name: str, variant: int = 0) -> Workload:
    return {"c2": workload_c2, "creal": workload_creal, "small": workload_small}[name](variant)


_GEOMETRY = {"c2": (64, 1875, 120.0, 40), "creal": (32, 192, 120.0, 40), "small": (16, 125, 25.0, 6)}  # rings, azimuths, half extent, boxes


def workload_scan_set(name: str, variant: int, k: int):
    """Scan set k > 0 of workload (name, variant): ANOTHER sweep against the same map -- another sensor pose along the street,
    other range noise -- with its ground truth and its perturbed guess, as the next scans of a sequence would be.  bench.py
    rotates through several sets per step so that consecutive batches do not re-align identical inputs.
    -> (scan_xyz [N,3] fp32, T_gt, T_guess)"""
    rings, azimuths, half_extent, n_boxes = _GEOMETRY[name]
    scene = make_scene(12345 + 1009 * variant, half_extent, n_boxes)
    rng = np.random.Generator(np.random.PCG64(9000 + 97 * variant + k))
    pose_gt = POSE_GT.copy()
    pose_gt[:2] += rng.uniform(-1.0, 1.0, 2) * np.array([0.05 * half_extent, 1.5])
    pose_gt[3] += rng.uniform(-0.3, 0.3)
    scan = make_scan(scene, pose_gt, rings, azimuths, 54321 + 31 * variant + 7919 * k)
    guess = pose_gt + GUESS_PERTURBATION
    return scan, pose_from_ypr(pose_gt), pose_from_ypr(guess)
