"""CPU oracle of the stand-alone odometry driver: the per-scan control flow of
mola::LidarOdometry::onLidarImpl (module/src/LidarOdometry.cpp:622-1313, relative to /root/reference), restated in
Python on top of the C oracle (oracle/icp_oracle.c) -- independently of the C++ driver under
mola_lidar_odometry_amd/host, including its own reading of the pipeline YAML (PyYAML + ${VAR|default} + formulas).

TEST INFRASTRUCTURE, PARITY UNPINNED (see oracle/icp_oracle.h): nothing under mola_lidar_odometry_amd/ imports this.
The motion model (mola::NavStateFuse [U]) and the key-frame list (mola::SearchablePoseList [U]) are the same plain
restatements the product documents in host/include/mola_lidar_odometry_hip/LidarOdometry.h.
"""
from __future__ import annotations

import math
import os
import re

import numpy as np
import yaml

from . import oracle_c as oc

_F = dict(max=max, min=min, sqrt=math.sqrt, abs=abs, sin=math.sin, cos=math.cos, tan=math.tan, exp=math.exp,
          log=math.log, pow=pow, pi=math.pi)


def _decimate_method(params):
    """decimate_method of a FilterDecimateVoxels block (default FirstPoint, lidar3d-default.yaml:291)."""
    m = str(params.get("decimate_method", "DecimateMethod::FirstPoint"))
    if m.endswith("FirstPoint"):
        return oc.DECIMATE_FIRST_POINT
    if m.endswith("ClosestToAverage"):
        return oc.DECIMATE_CLOSEST_TO_AVERAGE
    raise ValueError("unsupported decimate_method " + m)


def formula(expr, variables):
    """mp2p_icp::Parameterizable formulas (exprtk [U]): arithmetic, ^ as power, max/min/sqrt/..."""
    if isinstance(expr, (int, float)):
        return float(expr)
    s = str(expr).strip()
    if s.startswith("$f{") and s.endswith("}"):
        s = s[3:-1]
    return float(eval(s.replace("^", "**"), {"__builtins__": {}}, {**_F, **variables}))


def _subst_env(text):
    pat = re.compile(r"\$\{([A-Za-z_][A-Za-z0-9_]*)(?:\|((?:[^{}]|\{[^{}]*\})*))?\}")
    def rep(m):
        val = os.environ.get(m.group(1), m.group(2))
        return m.group(0) if val is None else val  # unknown and without default (only in comments): left alone
    for _ in range(8):  # defaults may contain further ${...}
        new = pat.sub(rep, text)
        if new == text:
            break
        text = new
    return text


def load_pipeline(path):
    with open(path) as f:
        return yaml.safe_load(_subst_env(f.read()))


def _b(v):
    return v if isinstance(v, bool) else str(v).strip().lower() in ("true", "1", "yes")


def _compose(A, B):
    return oc.pose_compose(A, B)


def _inv_compose(A, B):
    """A (-) B = B^-1 (+) A"""
    return oc.pose_compose(oc.pose_inverse(np.asarray(B, np.float64)), A)


def _rotvec_pose(w, v):
    xi = np.array([0, 0, 0, w[0], w[1], w[2]], np.float64)
    T = oc.se3_exp(xi).reshape(3, 4).copy()
    T[:, 3] = v
    return T.reshape(12)


def _bbox_radius(xyz):
    """max(|bb.max|, |bb.min|) with float norms (TPoint3Df::norm), LidarOdometry.cpp:1503-1508"""
    mn, mx = xyz.min(0).astype(np.float32), xyz.max(0).astype(np.float32)
    a = np.sqrt((mx[0] * mx[0] + mx[1] * mx[1]) + mx[2] * mx[2])
    b = np.sqrt((mn[0] * mn[0] + mn[1] * mn[1]) + mn[2] * mn[2])
    return float(max(a, b))


# Solver_GaussNewton.robustKernel (lidar3d-default.yaml:188); the forms beyond GemanMcClure are SURVEY App. B U1's candidates
_KERNELS = {"None": oc.KERNEL_NONE, "GemanMcClure": oc.KERNEL_GM_C4, "GemanMcClure_KISS": oc.KERNEL_GM_KISS,
            "GemanMcClure_Barron": oc.KERNEL_GM_BARRON, "Cauchy": oc.KERNEL_CAUCHY, "GemanMcClure_C2": oc.KERNEL_GM_C2}


class OdometryOracle:
    def __init__(self, pipeline_yaml_path, n_threads=8):
        c = load_pipeline(pipeline_yaml_path)
        self.cfg = c
        self.p = c["params"]
        self.n_threads = n_threads
        nav = c.get("navstate_fuse_params", {}) or {}
        self.max_time_vel = float(nav.get("max_time_to_use_velocity_model", 2.0))
        # the motion model's covariance as a prior of align() (LidarOdometry.cpp:859-861): covariance of the last fused
        # pose + (sigma_random_walk_acceleration * dt)^2, inverted [U]; `motion_model_prior: false` = no prior term
        self.sigma_lin = float(nav.get("sigma_random_walk_acceleration_linear", 1.0))
        self.sigma_ang = float(nav.get("sigma_random_walk_acceleration_angular", 10.0))
        self.motion_model_prior = _b(nav.get("motion_model_prior", False))
        it = [float(v) for v in (nav.get("initial_twist") or [])]
        self.initial_twist = np.array(it) if len(it) == 6 and any(v != 0.0 for v in it) else None
        f1 = c["observations_filter_1st_pass"]
        names = [e["class_name"].split("::")[-1] for e in f1]
        assert names == ["FilterDecimateVoxels", "FilterByRange", "FilterBoundingBox", "FilterDecimateVoxels"], names
        self.f1 = [e["params"] for e in f1]
        ts = c["observations_filter_adjust_timestamps"][0]["params"]
        self.ts_method = oc.TS_MIDDLE_IS_ZERO if str(ts["method"]).endswith("MiddleIsZero") else oc.TS_EARLIEST_IS_ZERO
        self.ts_offset = ts["time_offset"]
        icp = c["icp_settings_with_vel"]
        self.icp = icp
        self.matchers = icp["matchers"]
        self.solver = icp["solvers"][0]["params"]
        self.map_def = c["localmap_generator"][0]["params"]["metric_map_definition"]
        self.reset()

    def reset(self):
        self.vars = {}
        self.last_pose = np.eye(4)[:3].reshape(12).copy()
        self.nav_last = None       # (t, pose)
        self.nav_twist = None
        self.nav_cov = np.zeros((6, 6))
        self.last_mm = None        # (pose, twist)
        self.last_obs_tim = None
        self.last_icp_timestamp = None
        self.first_ever = None
        self.last_obs_timestamp = None
        self.sigma = 0.0
        self.est_range = None
        self.inst_range = None
        self.last_icp_quality = 0.0
        self.last_icp_was_good = True
        self.kfs = []
        self.removal_counter = 0
        self.map = None
        self.trajectory = []
        self.records = []

    # ---- motion model
    def _nav_reset(self):
        self.nav_last, self.nav_twist = None, None
        self.nav_cov = np.zeros((6, 6))

    def _nav_fuse(self, t, pose, cov=None):
        if self.nav_last is not None:
            dt = t - self.nav_last[0]
            if 0 < dt <= self.max_time_vel:
                inc = _inv_compose(pose, self.nav_last[1]).reshape(3, 4)
                w = oc.so3_log(inc.reshape(12))
                self.nav_twist = np.concatenate([inc[:, 3] / dt, w / dt])
            else:
                self.nav_twist = None
        elif self.initial_twist is not None:
            self.nav_twist = self.initial_twist.copy()
        self.nav_last = (t, np.array(pose, np.float64))
        self.nav_cov = np.array(cov, np.float64).reshape(6, 6) if cov is not None else np.eye(6) * 1e-12

    def _nav_estimate(self, t):
        """-> (pose, twist, prior information in the solver's tangent order [v; w], or None)"""
        if self.nav_last is None or self.nav_twist is None:
            return None
        dt = t - self.nav_last[0]
        if dt < 0 or dt > self.max_time_vel:
            return None
        tw = self.nav_twist
        info = None
        if self.motion_model_prior:
            perm = [0, 1, 2, 5, 4, 3]  # (x,y,z,yaw,pitch,roll) -> [v; w], w = (roll, pitch, yaw) to first order
            Cm = self.nav_cov[np.ix_(perm, perm)].copy()
            Cm[np.arange(3), np.arange(3)] += (self.sigma_lin * dt) ** 2
            Cm[np.arange(3, 6), np.arange(3, 6)] += (self.sigma_ang * dt) ** 2
            try:
                L = np.linalg.cholesky(Cm)
                Li = np.linalg.inv(L)
                info = Li.T @ Li
                if not np.all(np.isfinite(info)):
                    info = None
            except np.linalg.LinAlgError:
                info = None
        return _compose(self.nav_last[1], _rotvec_pose(tw[3:] * dt, tw[:3] * dt)), tw.copy(), info

    # ---- dynamic variables (LidarOdometry.cpp:1571-1635)
    def _update_vars(self):
        tw = self.last_mm[1] if self.last_mm is not None else np.zeros(6)
        for k, v in zip(("vx", "vy", "vz", "wx", "wy", "wz"), tw):
            self.vars[k] = float(v)
        ypr = oc.pose_to_ypr(self.last_pose)
        for k, v in zip(("robot_x", "robot_y", "robot_z", "robot_yaw", "robot_pitch", "robot_roll"), ypr):
            self.vars[k] = float(v)
        self.vars["ADAPTIVE_THRESHOLD_SIGMA"] = self.sigma if self.sigma != 0 else float(self.p["adaptive_threshold"]["initial_sigma"])
        self.vars["ICP_ITERATION"] = 0.0
        for k in ("icp_iterations", "SENSOR_TIME_OFFSET", "twistCorrectionCount"):
            self.vars.setdefault(k, 0.0)
        if self.est_range is not None:
            self.vars["ESTIMATED_SENSOR_MAX_RANGE"] = self.est_range
        self.vars["INSTANTANEOUS_SENSOR_MAX_RANGE"] = self.inst_range if self.inst_range is not None else 20.0

    def _deskew_layers(self):
        tw = [self.vars[k] for k in ("vx", "vy", "vz", "wx", "wy", "wz")]
        if self._t is None:
            self.for_map, self.for_icp = self._xyz[self.idx_map], self._xyz[self.idx_icp]
            return
        self.for_map = oc.deskew(self._xyz[self.idx_map], self._ta[self.idx_map], tw)
        self.for_icp = oc.deskew(self._xyz[self.idx_icp], self._ta[self.idx_icp], tw)

    def _schedules(self, n):
        thr, kp, pl = np.zeros(n), np.zeros(n), None
        m_pts = [m for m in self.matchers if m["class"].endswith("Matcher_Points_DistanceThreshold")][0]["params"]
        m_pl = [m for m in self.matchers if m["class"].endswith("Matcher_Point2Plane")]
        if m_pl:
            pl = np.zeros(n)
        for k in range(n):
            v = {**self.vars, "ICP_ITERATION": float(k)}
            thr[k] = formula(m_pts["threshold"], v)
            kp[k] = formula(self.solver["robustKernelParam"], v)
            if m_pl:
                pl[k] = formula(m_pl[0]["params"]["distanceThreshold"], v)
        return thr, kp, pl

    def on_lidar(self, stamp, xyz, t=None):
        P, A, L = self.p, self.p["adaptive_threshold"], self.p["local_map_updates"]
        xyz = np.ascontiguousarray(xyz, np.float32)
        rec = dict(timestamp=stamp, dropped=False, first_scan=False, icp_run=False, icp_good=False,
                   had_motion_model=False, map_updated=False, restarted=False, goodness=0.0, icp_iterations=0,
                   twist_corrections=0, align_calls=0, termination=0, n_raw=len(xyz))
        self.records.append(rec)
        if self.last_obs_tim is not None and stamp - self.last_obs_tim < float(P["min_time_between_scans"]):
            rec["dropped"] = True
            return rec
        if self.est_range is None and len(xyz):
            self.est_range = max(_bbox_radius(xyz[np.isfinite(xyz).all(1)]), float(P["absolute_minimum_sensor_range"]))
        self._update_vars()
        rec["twist"] = np.array([self.vars[k] for k in ("vx", "vy", "vz", "wx", "wy", "wz")])
        # ---- 1st pass + 2nd pass (yaml:278-350)
        v = self.vars
        d1, rg, bb, d2 = self.f1
        self.idx_map, self.idx_icp = oc.preprocess(
            xyz, formula(d1["voxel_filter_resolution"], v), formula(d2["voxel_filter_resolution"], v),
            int(d1["minimum_input_points_to_filter"]), oc.INDEX_FLOOR, formula(rg["range_min"], v),
            formula(rg["range_max"], v), (0, 0, 0), 1 if "outside_pointcloud_layer" in bb else 2,
            [formula(e, v) for e in bb["bounding_box_min"]], [formula(e, v) for e in bb["bounding_box_max"]],
            decim_map_method=_decimate_method(d1), decim_icp_method=_decimate_method(d2))
        self._xyz, self._t = xyz, t
        self._ta = None if t is None else oc.adjust_timestamps(t, self.ts_method, formula(self.ts_offset, v))
        self._deskew_layers()
        rec["n_for_map"], rec["n_for_icp"] = len(self.for_map), len(self.for_icp)
        rec["decim_map_resolution"] = formula(d1["voxel_filter_resolution"], v)
        rec["decim_icp_resolution"] = formula(d2["voxel_filter_resolution"], v)
        if self.est_range is not None:
            radius = max(_bbox_radius(self.for_icp) if len(self.for_icp) else 0.0, float(P["absolute_minimum_sensor_range"]))
            self.inst_range = radius
            a = float(P["max_sensor_range_filter_coefficient"])
            self.est_range = self.est_range * a + radius * (1.0 - a)
        rec["estimated_sensor_max_range"], rec["instantaneous_sensor_max_range"] = self.est_range, self.inst_range
        self.last_obs_tim = stamp
        self.last_obs_timestamp = stamp
        if self.first_ever is None:
            self.first_ever = stamp
        if len(xyz) == 0:
            rec["dropped"] = True
            return rec

        update_map = False
        self.last_mm = self._nav_estimate(stamp)
        has_mm = self.last_mm is not None
        rec["had_motion_model"] = has_mm
        if self.map is None or self.map.num_points == 0:
            rec["first_scan"] = True
            update_map = True
            self.trajectory.append((stamp, self.last_pose.copy()))
            self._nav_fuse(stamp, np.eye(4)[:3].reshape(12))
        else:
            guess_ypr = oc.pose_to_ypr(self.last_mm[0] if has_mm else self.last_pose)
            last_kf_pose = self.last_pose.copy()
            since_kf = (stamp - self.last_icp_timestamp) if self.last_icp_timestamp is not None else 0.0
            self.last_icp_timestamp = stamp
            cur = guess_ypr.copy()
            rec["init_guess"] = oc.pose_from_ypr(guess_ypr)
            ip = self.icp["params"]
            remaining = int(ip["maxIterations"])
            opt_twist = _b(P["optimize_twist"])
            while True:
                thr, kp, pl = self._schedules(remaining)
                T0 = oc.pose_from_ypr(cur)
                q = oc.ICPParams(max_iterations=remaining, min_abs_step_trans=float(ip["minAbsStep_trans"]),
                                 min_abs_step_rot=float(ip["minAbsStep_rot"]), threshold=thr, kernel_param=kp,
                                 pt2pl_threshold=pl, threshold_angular_deg=0.0,
                                 gn=oc.GNParams(max_inner_iterations=int(self.solver["maxIterations"]),
                                                robust_kernel=_KERNELS[str(self.solver.get("robustKernel", "GemanMcClure")).split("::")[-1]]),
                                 hook_enabled=opt_twist, hook_min_trans=float(P["optimize_twist_rerun_min_trans"]),
                                 hook_min_rot=math.radians(float(P["optimize_twist_rerun_min_rot_deg"])),
                                 hook_checkpoint=T0,
                                 # U12 (MOLA_HIP_MATCHED_POINTS=skip): points the plane matcher paired get no point pairing
                                 pt2pt_skip_plane_paired=os.environ.get("MOLA_HIP_MATCHED_POINTS", "again") in ("skip", "1"))
                prior = (self.last_mm[0], self.last_mm[2]) if (has_mm and self.last_mm[2] is not None) else None
                res = oc.icp_align(self.map, self.for_icp, T0, q, prior=prior, n_threads=self.n_threads)
                rec["align_calls"] += 1
                remaining -= min(remaining, res["n_iterations"])
                rec["icp_iterations"] += res["n_iterations"]
                if res["termination_reason"] != 6:  # HookRequest
                    break
                cur = oc.pose_to_ypr(res["T"])
                rec["twist_corrections"] += 1
                if since_kf > 0:
                    inc = _inv_compose(res["T"], last_kf_pose).reshape(3, 4)
                    w = oc.so3_log(inc.reshape(12))
                    tw = np.concatenate([inc[:, 3] / since_kf, w / since_kf])
                    for k, val in zip(("vx", "vy", "vz", "wx", "wy", "wz"), tw):
                        self.vars[k] = float(val)
                    self._deskew_layers()
                    rec["twist"] = tw
            rec["icp_run"] = True
            rec["termination"] = res["termination_reason"]
            rec["goodness"] = res["quality"]
            good = res["quality"] >= float(P["min_icp_goodness"])
            self.last_icp_was_good, self.last_icp_quality = good, res["quality"]
            rec["icp_good"] = good
            if good:
                self.last_pose = res["T"].copy()
                self._nav_fuse(stamp, res["T"], res["cov"])
                self.trajectory.append((stamp, self.last_pose.copy()))
            else:
                self._nav_reset()
            self.vars["icp_iterations"] = float(res["n_iterations"])
            if _b(A["enabled"]) and self.est_range is not None:  # :1052-1064, 1449-1485
                err = _inv_compose(res["T"], oc.pose_from_ypr(guess_ypr)).reshape(3, 4)
                theta = float(np.linalg.norm(oc.so3_log(err.reshape(12))))
                model_error = float(np.linalg.norm(err[:, 3])) + 2.0 * self.est_range * math.sin(theta / 2.0)
                rot_error = 0.1 * float(np.linalg.norm(self.last_mm[1][3:])) * self.est_range if has_mm else 0.0
                KP = float(A["kp"])
                new_sigma = (model_error + rot_error) * min(KP, max(0.1, KP * (1.0 - self.last_icp_quality)))
                if self.sigma == 0:
                    self.sigma = float(A["initial_sigma"])
                al = float(A["alpha"])
                self.sigma = al * self.sigma + (1.0 - al) * new_sigma
                self.sigma = min(float(A["maximum_sigma"]), max(float(A["min_motion"]), self.sigma))
            # key-frame decision (:1066-1118); the formulas see the variables of this scan's last realize()
            if self.kfs:
                d = [float(np.sum((k.reshape(3, 4)[:, 3] - self.last_pose.reshape(3, 4)[:, 3]) ** 2)) for k in self.kfs]
                rel = _inv_compose(self.last_pose, self.kfs[int(np.argmin(d))]).reshape(3, 4)
                first, dist, rot = False, float(np.linalg.norm(rel[:, 3])), float(np.linalg.norm(oc.so3_log(rel.reshape(12))))
            else:
                first, dist, rot = True, 0.0, 0.0
            update_map = (good and _b(L["enabled"]) and has_mm and
                          (first or dist > formula(L["min_translation_between_keyframes"], self.vars) or
                           rot > math.radians(formula(L["min_rotation_between_keyframes"], self.vars))))
            if update_map:
                self.kfs.append(self.last_pose.copy())
                maxd = formula(L["max_distance_to_keep_keyframes"], self.vars)
                if maxd > 0:
                    c = self.removal_counter
                    self.removal_counter += 1
                    if c >= int(L["check_for_removal_every_n"]):
                        self.removal_counter = 0
                        tp = self.last_pose.reshape(3, 4)[:, 3]
                        self.kfs = [k for k in self.kfs if np.linalg.norm(k.reshape(3, 4)[:, 3] - tp) <= maxd]
        if (not self.last_icp_was_good) and len(self.trajectory) == 1:
            self.map = oc.Map(*self._map_args) if self.map is not None else None  # local_map->clear()
            self.trajectory = []
            update_map = False
            self.last_icp_was_good = True
            rec["restarted"] = True
        if update_map:
            if self.map is None:
                co, io = self.map_def["creationOpts"], self.map_def["insertOpts"]
                self.voxel_size = formula(co["voxel_size"], self.vars)
                ndt = self.map_def["class"].endswith("NDT")
                self._map_args = (np.float32(self.voxel_size), int(formula(io["max_points_per_voxel"], self.vars)), oc.INDEX_FLOOR,
                                  float(io.get("min_distance_between_points", 0.0)),
                                  float(io.get("max_eigen_ratio_for_planes", 0.05)) if ndt else 0.0, 4)
                self.map = oc.Map(*self._map_args)
                self.remove_far = float(np.float32(formula(io.get("remove_voxels_farther_than", 0.0), self.vars)))
            self._update_vars()
            self.map.insert_posed(self.for_map, self.last_pose, self.remove_far)
            rec["map_updated"] = True
        rec["pose"] = self.last_pose.copy()
        rec["sigma"] = self.sigma
        rec["n_map_points"] = self.map.num_points if self.map is not None else 0
        rec["n_map_voxels"] = self.map.num_voxels if self.map is not None else 0
        return rec
