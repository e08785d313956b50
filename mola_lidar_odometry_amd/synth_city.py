"""A synthetic CITY drive for the odometry-level measurements (bench.py `single_sequence*`, tests/test_gpu_accuracy.py).

No dataset exists in either container, and the street canyon of synth.py is too poor for an accuracy
statement: two facades and a ground plane leave the along-street motion weakly observable, the ICP layer ends up with
~900 points where a KITTI scan gives 3-8 k, and a 100-scan drive never fills the local map.  This module
generates what VERDICT r3 asked for instead: a grid of streets with cross streets, buildings of varying set-back, width,
height and gaps, parks, parked cars, poles and trees (boxes + capped cylinders on the ground plane), and a route of any
length through it -- straights, left and right turns, speed changes -- with exact ground truth.  The sweeps are cast by
synthgen/city_raycast.c (grid-accelerated, OpenMP): ~120 k returns of an HDL-64-like sensor (64 rings x 1875 azimuths,
+2 .. -24.8 deg, 80 m effective range as on KITTI) per sweep, skewed by the vehicle's motion during the sweep.

Data generation only: not the product path, not the oracle.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import synth

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmolahip_synth.so")
        if not os.path.exists(path):
            raise RuntimeError("libmolahip_synth.so is missing: run __graft_entry__.build() (make -C mola_lidar_odometry_amd/synthgen)")
        L = ctypes.CDLL(path)
        L.city_create.restype = ctypes.c_void_p
        L.city_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_double] * 5
        L.city_destroy.argtypes = [ctypes.c_void_p]
        L.city_sweep.restype = ctypes.c_int
        L.city_sweep.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_double, ctypes.c_double, ctypes.c_uint64, ctypes.c_double, ctypes.c_double,
                                 ctypes.c_void_p, ctypes.c_void_p]
        L.city_set_threads.argtypes = [ctypes.c_int]
        L.city_set_threads(max(1, min(16, os.cpu_count() or 1)))
        _LIB = L
    return _LIB


PITCH = 90.0          # street grid pitch [m]
ROAD_HALF = 4.0       # carriageway half width
FACADE = 10.0         # nearest facade line, from the street's centre line
GRID_X = (-2, 8)      # intersections i = -2 .. 8  (x = i * PITCH)
GRID_Y = (-2, 7)


def make_city(seed: int = 2024):
    """-> dict(boxes [B,6] f32, cyls [C,5] f32, bounds (xmin, ymin, xmax, ymax)).  Blocks between the streets are built up
    (50 %), parks (30 %) or open lots with parked cars (20 %)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    boxes, cyls = [], []

    def tree(x, y):
        # trunk + a crown of foliage clumps: a LiDAR sees INTO a crown (returns from its whole volume), which is where a
        # real scan's occupied voxels come from -- a solid crown would only show its skin
        h = rng.uniform(2.2, 4.0)
        cyls.append([x, y, rng.uniform(0.12, 0.3), 0.0, h])
        rad, top = rng.uniform(1.6, 3.4), h + rng.uniform(2.5, 6.0)
        cyls.append([x, y, 0.45 * rad, h, top - 0.5])  # the dense core
        for _ in range(int(rng.integers(14, 30))):
            a, rr, zz = rng.uniform(0, 2 * np.pi), rad * np.sqrt(rng.uniform()), rng.uniform(h - 0.5, top)
            s_ = rng.uniform(0.15, 0.45)
            boxes.append([x + rr * np.cos(a) - s_, y + rr * np.sin(a) - s_, zz - s_, x + rr * np.cos(a) + s_, y + rr * np.sin(a) + s_, zz + s_])

    def car(x, y, along_x):
        l, w, h = rng.uniform(3.8, 4.8), rng.uniform(1.6, 1.9), rng.uniform(1.35, 1.9)
        sx, sy = (l, w) if along_x else (w, l)
        boxes.append([x - sx / 2, y - sy / 2, 0.0, x + sx / 2, y + sy / 2, h])

    for i in range(GRID_X[0], GRID_X[1]):
        for j in range(GRID_Y[0], GRID_Y[1]):
            x0, x1 = i * PITCH + FACADE, (i + 1) * PITCH - FACADE
            y0, y1 = j * PITCH + FACADE, (j + 1) * PITCH - FACADE
            kind = rng.uniform()
            if kind < 0.50:  # built-up, suburban: two rows of detached houses with gaps, hedges in front, yards with trees
                for row, (inset, gap_lo, gap_hi) in enumerate(((0.0, 3.5, 10.0), (20.0, 5.0, 14.0))):
                    for side in range(4):
                        a0, a1 = ((x0, x1) if side < 2 else (y0, y1))
                        a0, a1 = a0 + inset, a1 - inset
                        a = a0 + rng.uniform(0.0, 4.0)
                        while a < a1 - 6.0:
                            wdt = min(rng.uniform(7.0, 15.0), a1 - a)
                            dep, hgt = rng.uniform(7.0, 11.0), rng.uniform(3.5, 11.0)
                            setb = inset + rng.uniform(2.0, 7.0)
                            if side == 0:
                                boxes.append([a, y0 + setb, 0.0, a + wdt, y0 + setb + dep, hgt])
                            elif side == 1:
                                boxes.append([a, y1 - setb - dep, 0.0, a + wdt, y1 - setb, hgt])
                            elif side == 2:
                                boxes.append([x0 + setb, a, 0.0, x0 + setb + dep, a + wdt, hgt])
                            else:
                                boxes.append([x1 - setb - dep, a, 0.0, x1 - setb, a + wdt, hgt])
                            if row == 0 and rng.uniform() < 0.25:  # hedge / fence along the pavement: the sensor looks over it
                                hh, th = rng.uniform(0.7, 1.4), rng.uniform(0.15, 0.5)
                                if side == 0:
                                    boxes.append([a, y0 + 0.3, 0.0, a + wdt, y0 + 0.3 + th, hh])
                                elif side == 1:
                                    boxes.append([a, y1 - 0.3 - th, 0.0, a + wdt, y1 - 0.3, hh])
                                elif side == 2:
                                    boxes.append([x0 + 0.3, a, 0.0, x0 + 0.3 + th, a + wdt, hh])
                                else:
                                    boxes.append([x1 - 0.3 - th, a, 0.0, x1 - 0.3, a + wdt, hh])
                            a += wdt + rng.uniform(gap_lo, gap_hi)
                for _ in range(rng.integers(14, 26)):  # yards: trees, sheds, bushes
                    px, py = rng.uniform(x0 + 2, x1 - 2), rng.uniform(y0 + 2, y1 - 2)
                    u = rng.uniform()
                    if u < 0.5:
                        tree(px, py)
                    elif u < 0.75:
                        cyls.append([px, py, rng.uniform(0.5, 1.3), 0.0, rng.uniform(0.8, 2.2)])  # bush
                    else:
                        s_ = rng.uniform(1.2, 3.0)
                        boxes.append([px - s_, py - s_, 0.0, px + s_, py + s_, rng.uniform(2.0, 3.2)])
            elif kind < 0.80:  # park
                for _ in range(rng.integers(25, 45)):
                    tree(rng.uniform(x0, x1), rng.uniform(y0, y1))
                for _ in range(rng.integers(3, 8)):  # benches / kiosks
                    px, py, s = rng.uniform(x0 + 3, x1 - 3), rng.uniform(y0 + 3, y1 - 3), rng.uniform(0.6, 2.5)
                    boxes.append([px - s, py - s / 2, 0.0, px + s, py + s / 2, rng.uniform(0.5, 3.0)])
            else:  # open lot with parked cars in rows
                for ry in np.arange(y0 + 4, y1 - 4, 7.5):
                    for rx in np.arange(x0 + 3, x1 - 3, 2.9):
                        if rng.uniform() < 0.55:
                            car(rx, ry, False)
                for _ in range(6):
                    px, py = rng.uniform(x0, x1), rng.uniform(y0, y1)
                    cyls.append([px, py, rng.uniform(0.08, 0.15), 0.0, rng.uniform(6.0, 9.0)])
    # street furniture along every street segment: poles, trees on the pavements, parked cars at the kerb
    for horizontal in (True, False):
        n_lines = range(GRID_Y[0], GRID_Y[1] + 1) if horizontal else range(GRID_X[0], GRID_X[1] + 1)
        n_segs = range(GRID_X[0], GRID_X[1]) if horizontal else range(GRID_Y[0], GRID_Y[1])
        for ln in n_lines:
            c = ln * PITCH
            for sg in n_segs:
                a0, a1 = sg * PITCH + FACADE + 2.0, (sg + 1) * PITCH - FACADE - 2.0
                for side in (-1.0, 1.0):
                    a = a0 + rng.uniform(0.0, 12.0)
                    while a < a1:
                        off = side * rng.uniform(5.4, 6.2)
                        x, y = (a, c + off) if horizontal else (c + off, a)
                        if rng.uniform() < 0.45:
                            tree(x, y)
                        else:
                            cyls.append([x, y, rng.uniform(0.08, 0.18), 0.0, rng.uniform(5.0, 9.0)])
                        a += rng.uniform(9.0, 26.0)
                    a = a0 + rng.uniform(0.0, 6.0)
                    while a < a1 - 5.0:
                        if rng.uniform() < 0.4:
                            off = side * rng.uniform(2.9, 3.2)
                            x, y = (a + 2.4, c + off) if horizontal else (c + off, a + 2.4)
                            car(x, y, horizontal)
                        a += rng.uniform(5.6, 7.5)
    bounds = (GRID_X[0] * PITCH - 30.0, GRID_Y[0] * PITCH - 30.0, GRID_X[1] * PITCH + 30.0, GRID_Y[1] * PITCH + 30.0)
    return dict(boxes=np.asarray(boxes, dtype=np.float32), cyls=np.asarray(cyls, dtype=np.float32), bounds=bounds, seed=seed)


class Raycaster:
    """ctypes handle on synthgen/city_raycast.c for one city."""

    def __init__(self, city, cell: float = 8.0):
        self.city = city
        b, c = np.ascontiguousarray(city["boxes"], np.float32), np.ascontiguousarray(city["cyls"], np.float32)
        self._keep = (b, c)
        self.h = _lib().city_create(b.ctypes.data, len(b), c.ctypes.data, len(c), *[float(v) for v in city["bounds"]], float(cell))

    def sweep(self, T_ref, twist, sweep_time=0.1, rings=64, azimuths=1875, seed=1, max_range=80.0, range_noise=0.02,
              el_top_deg=2.0, el_bottom_deg=-24.8):
        T = np.ascontiguousarray(np.asarray(T_ref, np.float64).reshape(12))
        tw = np.ascontiguousarray(np.asarray(twist, np.float64).reshape(6))
        xyz = np.empty((rings * azimuths, 3), np.float32)
        t = np.empty(rings * azimuths, np.float32)
        n = _lib().city_sweep(self.h, T.ctypes.data, tw.ctypes.data, float(sweep_time), int(rings), int(azimuths), float(el_top_deg),
                              float(el_bottom_deg), int(seed), float(max_range), float(range_noise), xyz.ctypes.data, t.ctypes.data)
        return xyz[:n].copy(), t[:n].copy()

    def close(self):
        if self.h:
            _lib().city_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


# the route: intersections (i, j) of the street grid, in order; left and right turns, 18 blocks = 1620 m
ROUTE = [(0, 0), (3, 0), (3, 1), (5, 1), (5, 4), (2, 4), (2, 2), (0, 2), (0, 0), (3, 0)]


def _planned_path(ds=0.1, radius=9.0):
    """Centre-line path through ROUTE with circular arcs at the corners: arrays x, y, heading, curvature sampled every ds."""
    P = [np.array([i * PITCH, j * PITCH], float) for i, j in ROUTE]
    xs, ys, hs, ks = [], [], [], []
    pos = P[0].copy()
    for k in range(1, len(P)):
        d_in = (P[k] - P[k - 1]) / np.linalg.norm(P[k] - P[k - 1])
        last = k == len(P) - 1
        end = P[k] - d_in * (0.0 if last else radius)
        L = np.dot(end - pos, d_in)
        n = max(1, int(round(L / ds)))
        for q in range(n):
            p = pos + d_in * (L * q / n)
            xs.append(p[0]), ys.append(p[1]), hs.append(np.arctan2(d_in[1], d_in[0])), ks.append(0.0)
        pos = end
        if last:
            break
        d_out = (P[k + 1] - P[k]) / np.linalg.norm(P[k + 1] - P[k])
        turn = np.sign(d_in[0] * d_out[1] - d_in[1] * d_out[0])  # +1 left
        centre = pos + turn * radius * np.array([-d_in[1], d_in[0]])
        h0 = np.arctan2(d_in[1], d_in[0])
        n = max(1, int(round(radius * (np.pi / 2) / ds)))
        for q in range(n):
            a = (np.pi / 2) * q / n
            h = h0 + turn * a
            p = centre + radius * np.array([np.sin(h) * turn, -np.cos(h) * turn])
            xs.append(p[0]), ys.append(p[1]), hs.append(h), ks.append(turn / radius)
        pos = P[k] + d_out * radius
    return np.asarray(xs), np.asarray(ys), np.unwrap(np.asarray(hs)), np.asarray(ks)


def route_plan(n_scans: int = 1000, dt: float = 0.1, v_max: float = 9.0, v_turn: float = 4.0, accel: float = 1.5,
               brake: float = 2.0, seed: int = 2024, lookahead: float = 5.0, v_start: float = 0.25):
    """Ground-truth poses (12 doubles), per-scan twists, stamps and sweep seeds.  pose[k+1] = pose[k] (+) (Exp_SO3(w dt),
    v dt) with the twist of scan k -- the recurrence of synth.drive_plan, so a constant-velocity model is exact up to the
    twist's variation -- steered along the planned path by pure pursuit.  The vehicle pulls away from rest."""
    ds = 0.1
    px, py, ph, pk = _planned_path(ds)
    n = len(px)
    # speed limit by curvature (smoothed over 15 m so that braking starts before the corner), then accel / brake passes
    lim = np.where(np.abs(pk) > 0, v_turn, v_max)
    w = int(8.0 / ds)
    lim = np.minimum.reduce([np.roll(lim, s) for s in range(-w, w + 1, max(1, w // 8))])
    v = lim.copy()
    v[0] = v_start
    for i in range(1, n):
        v[i] = min(v[i], np.sqrt(v[i - 1] ** 2 + 2 * accel * ds))
    for i in range(n - 2, -1, -1):
        v[i] = min(v[i], np.sqrt(v[i + 1] ** 2 + 2 * brake * ds))
    T = np.zeros((3, 4))
    c0, s0 = np.cos(ph[0]), np.sin(ph[0])
    T[:, :3] = [[c0, -s0, 0], [s0, c0, 0], [0, 0, 1]]
    T[:, 3] = [px[0], py[0] + 0.4, synth.SENSOR_H]
    poses, twists, stamps, seeds = [], [], [], []
    idx = 0
    for k in range(n_scans):
        # nearest path sample ahead of the last one
        win = slice(idx, min(n, idx + 400))
        idx = idx + int(np.argmin((px[win] - T[0, 3]) ** 2 + (py[win] - T[1, 3]) ** 2))
        if idx >= n - int(lookahead / ds) - 2:
            raise ValueError("route_plan: the route is shorter than %d scans (reached its end at scan %d)" % (n_scans, k))
        speed = float(v[idx]) * (1.0 + 0.03 * np.sin(0.7 * k))
        tgt = min(n - 1, idx + int(lookahead / ds))
        dx, dy = px[tgt] - T[0, 3], py[tgt] - T[1, 3]
        lx, ly = T[0, 0] * dx + T[1, 0] * dy, T[0, 1] * dx + T[1, 1] * dy  # look-ahead point in the vehicle frame
        wz = 2.0 * speed * ly / max(1e-6, lx * lx + ly * ly)
        # suspension: the body pitches, rolls and heaves a little (sums of incommensurate sinusoids, followed with a weak
        # restoring feedback so that the attitude stays bounded).  Without it the ground rings of consecutive sweeps over
        # a flat ground coincide, and point-to-point pairings between them pull every estimate towards "no motion".
        tk = k * dt
        sway = min(1.0, speed / 2.0)
        pitch_t = sway * (0.0040 * np.sin(2 * np.pi * 0.83 * tk) + 0.0020 * np.sin(2 * np.pi * 2.3 * tk + 1.0))
        roll_t = sway * (0.0035 * np.sin(2 * np.pi * 0.61 * tk + 0.5) + 0.0015 * np.sin(2 * np.pi * 1.9 * tk))
        z_t = synth.SENSOR_H + sway * (0.015 * np.sin(2 * np.pi * 1.3 * tk + 0.3) + 0.008 * np.sin(2 * np.pi * 2.9 * tk))
        pitch_a, roll_a = -np.arcsin(np.clip(T[2, 0], -1, 1)), np.arctan2(T[2, 1], T[2, 2])
        wy = (pitch_t - pitch_a) / dt * 0.8
        wx = (roll_t - roll_a) / dt * 0.8
        vz = (z_t - T[2, 3]) / dt * 0.8
        tw = np.array([speed, 0.03 * np.cos(0.5 * k) * min(1.0, speed), vz, wx, wy, wz])
        poses.append(T.reshape(12).copy())
        twists.append(tw)
        stamps.append(1000.0 + k * dt)
        seeds.append(seed + 10 * k)
        Rn = synth._so3_exp((tw[3:] * dt)[None])[0]
        T = np.concatenate([T[:, :3] @ Rn, (T[:, :3] @ (tw[:3] * dt) + T[:, 3])[:, None]], axis=1)
    return dict(poses=np.asarray(poses), twists=np.asarray(twists), stamps=np.asarray(stamps), seeds=seeds, dt=dt)


def path_length(poses) -> float:
    p = np.asarray(poses).reshape(-1, 3, 4)[:, :, 3]
    return float(np.linalg.norm(np.diff(p, axis=0), axis=1).sum())


def make_city_drive(n_scans: int = 1000, rings: int = 64, azimuths: int = 1875, seed: int = 2024, max_range: float = 80.0,
                    range_noise: float = 0.02, dt: float = 0.1, caster: Raycaster | None = None, sink=None, skew: bool = True,
                    v_start: float = 0.25):
    """The drive: dict(poses, twists, stamps, scans=[(xyz, t)]) like synth.make_drive.  `sink(k, xyz, t)`: called per
    sweep instead of keeping the scans in memory (a 1000-scan drive is 1.9 GB).  skew=False: the sensor stands still during
    a sweep (what a motion-compensated data set such as KITTI's velodyne folder holds)."""
    rc = caster or Raycaster(make_city(seed))
    plan = route_plan(n_scans, dt=dt, seed=seed, v_start=v_start)
    scans = []
    for k in range(n_scans):
        xyz, t = rc.sweep(plan["poses"][k], plan["twists"][k] if skew else np.zeros(6), dt, rings, azimuths, plan["seeds"][k], max_range,
                          range_noise)
        if sink is not None:
            sink(k, xyz, t)
        else:
            scans.append((xyz, t))
    return dict(poses=plan["poses"], twists=plan["twists"], stamps=plan["stamps"], scans=scans, city=rc.city)


def write_kitti_drive(root: str, n_scans: int = 1000, seq: str = "00", time_channel: bool = True, **kw) -> tuple[str, dict]:
    """Cast the drive straight into a KITTI odometry sequence folder (velodyne/%06d.bin rows of four float32 + times.txt).
    The fourth float of a row is the intensity in KITTI; with time_channel it carries the point's time stamp relative to
    the scan's [s] (molahip-lo-cli --time-field 12), so that the sweeps can stay skewed and the de-skew filter has work.
    -> (sequence directory, drive without the scans, with points_per_scan)."""
    d = os.path.join(root, "sequences", seq)
    os.makedirs(os.path.join(d, "velodyne"), exist_ok=True)
    sizes = []

    def sink(k, xyz, t):
        rows = np.zeros((len(xyz), 4), np.float32)
        rows[:, :3] = xyz
        if time_channel:
            rows[:, 3] = t
        rows.tofile(os.path.join(d, "velodyne", "%06d.bin" % k))
        sizes.append(len(xyz))

    drive = make_city_drive(n_scans, sink=sink, **kw)
    np.savetxt(os.path.join(d, "times.txt"), drive["stamps"] - drive["stamps"][0], fmt="%.6e")
    drive["points_per_scan"] = sizes
    return d, drive


def ground_truth_44(drive) -> np.ndarray:
    """[N,4,4] ground-truth poses relative to the first one (the odometry's frame)."""
    n = len(drive["poses"])
    gt = np.tile(np.eye(4), (n, 1, 1))
    gt[:, :3, :] = np.asarray(drive["poses"]).reshape(n, 3, 4)
    return np.einsum("ij,njk->nik", np.linalg.inv(gt[0]), gt)
