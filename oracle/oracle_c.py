"""ctypes binding of the C oracle (oracle/icp_oracle.c).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under mola_lidar_odometry_amd/ may import this module.
PARITY UNPINNED -- see oracle/icp_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libicp_oracle.so")

KERNEL_NONE, KERNEL_GM_C4, KERNEL_GM_KISS, KERNEL_GM_BARRON, KERNEL_CAUCHY, KERNEL_GM_C2 = range(6)
INDEX_FLOOR, INDEX_TRUNC = 0, 1
FAR_CHEBYSHEV, FAR_L1, FAR_L2 = 0, 1, 2
PT2PL_PLANE_DISTANCE, PT2PL_CENTROID_DISTANCE = 0, 1  # the oracle takes the second as a NEGATIVE distance threshold
TERM_NAMES = ["Undefined", "NoPairings", "SolverError", "MaxIterations", "Stalled",
              "QualityCheckpointFailed", "HookRequest"]


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("icp_oracle.c", "icp_oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


class _MapParams(C.Structure):
    _fields_ = [("voxel_size", C.c_float), ("max_points_per_voxel", C.c_uint32), ("index_mode", C.c_uint32),
                ("min_distance_between_points", C.c_float), ("ndt_max_eigen_ratio", C.c_float),
                ("ndt_min_points", C.c_uint32), ("far_voxel_metric", C.c_uint32)]


class _PreprocessParams(C.Structure):
    _fields_ = [("decim_map_resolution", C.c_float), ("decim_icp_resolution", C.c_float),
                ("min_points_to_filter", C.c_uint32), ("index_mode", C.c_int32), ("range_min", C.c_float),
                ("range_max", C.c_float), ("range_center", C.c_float * 3), ("bbox_mode", C.c_int32),
                ("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3), ("decim_map_method", C.c_int32),
                ("decim_icp_method", C.c_int32)]


DECIMATE_FIRST_POINT, DECIMATE_CLOSEST_TO_AVERAGE = 0, 1


class _MatchStats(C.Structure):
    _fields_ = [("potential_pairings", C.c_uint64), ("n_candidates", C.c_uint64), ("n_voxels_hit", C.c_uint64)]


_FP = C.POINTER(C.c_float)
_UP = C.POINTER(C.c_uint32)
_DP = C.POINTER(C.c_double)


class _PairsPt2Pt(C.Structure):
    _fields_ = [("lx", _FP), ("ly", _FP), ("lz", _FP), ("gx", _FP), ("gy", _FP), ("gz", _FP), ("n", C.c_size_t)]


class _PairsPt2Pl(C.Structure):
    _fields_ = [("lx", _FP), ("ly", _FP), ("lz", _FP), ("cx", _FP), ("cy", _FP), ("cz", _FP),
                ("nx", _FP), ("ny", _FP), ("nz", _FP), ("n", C.c_size_t)]


class _Prior(C.Structure):
    _fields_ = [("mean", C.c_double * 12), ("info", C.c_double * 36)]


class _GNParams(C.Structure):
    _fields_ = [("max_inner_iterations", C.c_uint32), ("robust_kernel", C.c_uint32),
                ("robust_kernel_param", C.c_double), ("min_delta", C.c_double), ("max_cost", C.c_double),
                ("weight_pt2pt", C.c_double), ("weight_pt2pl", C.c_double)]


class _GNStep(C.Structure):
    _fields_ = [("H", C.c_double * 36), ("g", C.c_double * 6), ("err_norm_sqr", C.c_double),
                ("delta", C.c_double * 6), ("T_after", C.c_double * 12)]


class _ICPParams(C.Structure):
    _fields_ = [("max_iterations", C.c_uint32), ("min_abs_step_trans", C.c_double), ("min_abs_step_rot", C.c_double),
                ("disable_stall_test", C.c_uint32), ("threshold", _DP), ("threshold_angular_deg", C.c_double),
                ("pt2pl_threshold", _DP), ("kernel_param", _DP), ("gn", _GNParams), ("hook_enabled", C.c_uint32),
                ("hook_min_trans", C.c_double), ("hook_min_rot", C.c_double), ("hook_checkpoint", C.c_double * 12),
                ("compute_covariance", C.c_uint32), ("cov_findif_xyz", C.c_double), ("cov_findif_ang", C.c_double),
                ("pt2pt_skip_plane_paired", C.c_uint32)]


class _ICPIter(C.Structure):
    _fields_ = [("T", C.c_double * 12), ("n_pairs", C.c_uint32), ("threshold", C.c_double),
                ("kernel_param", C.c_double), ("delta_trans", C.c_double), ("delta_rot", C.c_double)]


class _ICPResult(C.Structure):
    _fields_ = [("T", C.c_double * 12), ("cov", C.c_double * 36), ("quality", C.c_double),
                ("n_iterations", C.c_uint32), ("termination_reason", C.c_uint32), ("n_final_pairs", C.c_uint32),
                ("potential_pairings", C.c_uint64), ("n_candidates_total", C.c_uint64),
                ("n_final_pairs_pt2pl", C.c_uint32)]


class _PairsOut(C.Structure):
    _fields_ = [("local_idx", _UP), ("global_idx", _UP), ("gx", _FP), ("gy", _FP), ("gz", _FP), ("d2", _FP)]


class _Pt2PlKnnParams(C.Structure):
    _fields_ = [("distance_threshold", C.c_double), ("plane_eigen_threshold", C.c_double), ("search_radius", C.c_double),
                ("knn", C.c_uint32), ("minimum_plane_points", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_map_create.restype = C.c_void_p
        L.orc_map_create.argtypes = [C.POINTER(_MapParams)]
        L.orc_map_destroy.argtypes = [C.c_void_p]
        L.orc_map_insert.argtypes = [C.c_void_p, _FP, _FP, _FP, C.c_size_t]
        L.orc_map_num_points.restype = C.c_size_t
        L.orc_map_num_points.argtypes = [C.c_void_p]
        L.orc_map_num_voxels.restype = C.c_size_t
        L.orc_map_num_voxels.argtypes = [C.c_void_p]
        L.orc_map_bbox.argtypes = [C.c_void_p, _FP, _FP]
        L.orc_map_dump.argtypes = [C.c_void_p, _FP, _FP, _FP, _UP, C.POINTER(C.c_int32), _UP, _UP]
        L.orc_map_nn_single.restype = C.c_int
        L.orc_map_nn_single.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, _FP, _FP, _UP,
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_match_points.restype = C.c_size_t
        L.orc_match_points.argtypes = [C.c_void_p, _FP, _FP, _FP, C.c_size_t, _DP, C.c_double, C.c_double,
                                       _UP, _UP, _FP, _FP, _FP, _FP, C.POINTER(_MatchStats), C.c_int]
        L.orc_match_pt2pl.restype = C.c_size_t
        L.orc_match_pt2pl.argtypes = [C.c_void_p, _FP, _FP, _FP, C.c_size_t, _DP, C.c_double, _UP, _FP, _FP, _FP, _FP, _FP,
                                      _FP, C.c_int]
        L.orc_map_dump_ndt.argtypes = [C.c_void_p, _FP, _FP, _FP, _FP, _FP, _FP, _UP]
        L.orc_match_pt2pl_knn.restype = C.c_size_t
        L.orc_match_pt2pl_knn.argtypes = [C.c_void_p, _FP, _FP, _FP, C.c_size_t, _DP, C.POINTER(_Pt2PlKnnParams), _UP] + [_FP] * 6
        L.orc_gn_solve.restype = C.c_int
        L.orc_gn_solve.argtypes = [C.POINTER(_PairsPt2Pt), C.POINTER(_PairsPt2Pl), C.POINTER(_GNParams),
                                   C.POINTER(_Prior), _DP, C.POINTER(_GNStep), C.c_int]
        L.orc_covariance.argtypes = [C.POINTER(_PairsPt2Pt), C.POINTER(_PairsPt2Pl), _DP, C.c_double, C.c_double,
                                     _DP, _DP]
        L.orc_icp_align.restype = C.c_int
        L.orc_icp_align.argtypes = [C.c_void_p, _FP, _FP, _FP, C.c_size_t, _DP, C.POINTER(_ICPParams),
                                    C.POINTER(_Prior), C.POINTER(_ICPResult), C.POINTER(_ICPIter),
                                    C.POINTER(_PairsOut), C.c_int]
        L.orc_max_threads.restype = C.c_int
        L.orc_map_insert_posed.argtypes = [C.c_void_p, _FP, _FP, _FP, C.c_size_t, _DP, C.c_float]
        L.orc_adjust_timestamps.argtypes = [_FP, C.c_size_t, C.c_int, C.c_float]
        L.orc_decimate_first_point.restype = C.c_size_t
        L.orc_decimate_first_point.argtypes = [_FP, _FP, _FP, C.c_size_t, C.c_float, C.c_uint32, C.c_int, _UP]
        L.orc_filter_by_range.restype = C.c_size_t
        L.orc_filter_by_range.argtypes = [_FP, _FP, _FP, C.c_size_t, C.c_float, C.c_float, _FP, _UP]
        L.orc_filter_bbox.restype = C.c_size_t
        L.orc_filter_bbox.argtypes = [_FP, _FP, _FP, C.c_size_t, _FP, _FP, C.c_int, _UP]
        L.orc_deskew.argtypes = [_FP, _FP, _FP, _FP, C.c_size_t, _DP, _FP, _FP, _FP]
        L.orc_decimate_closest_to_average.restype = C.c_size_t
        L.orc_decimate_closest_to_average.argtypes = [_FP, _FP, _FP, C.c_size_t, C.c_float, C.c_uint32, C.c_int, _UP]
        L.orc_preprocess.argtypes = [_FP, _FP, _FP, C.c_size_t, C.POINTER(_PreprocessParams), _UP,
                                     C.POINTER(C.c_size_t), _UP, C.POINTER(C.c_size_t)]
        for name in ("orc_pose_from_ypr", "orc_pose_to_ypr", "orc_se3_exp", "orc_se3_log", "orc_pose_inverse"):
            getattr(L, name).argtypes = [_DP, _DP]
        L.orc_so3_log.argtypes = [_DP, _DP]
        L.orc_pose_compose.argtypes = [_DP, _DP, _DP]
        _lib = _TimedLib(L)
    return _lib


# Seconds spent INSIDE the C library by the heavy calls (matching, solving, filters, map insertion) since reset_c_seconds():
# lets a Python loop over the oracle (odometry_oracle.py) report its C share, so that a CPU scans/s figure can be quoted
# without the interpreter's overhead ("time the CPU drivers through a C/C++ loop or report the Python share").
C_SECONDS = 0.0
_TIMED = {"orc_icp_align", "orc_preprocess", "orc_deskew", "orc_map_insert_posed", "orc_map_insert", "orc_match_points",
          "orc_match_pt2pl", "orc_gn_solve", "orc_covariance", "orc_decimate_first_point", "orc_filter_by_range",
          "orc_filter_bbox", "orc_adjust_timestamps"}


def reset_c_seconds():
    global C_SECONDS
    C_SECONDS = 0.0


class _TimedLib:
    def __init__(self, L):
        self._L = L
        self._cache = {}

    def __getattr__(self, name):
        f = self._cache.get(name)
        if f is None:
            raw = getattr(self._L, name)
            if name in _TIMED:
                import time as _time

                def f(*a, _raw=raw, _pc=_time.perf_counter):
                    global C_SECONDS
                    t0 = _pc()
                    try:
                        return _raw(*a)
                    finally:
                        C_SECONDS += _pc() - t0
            else:
                f = raw
            self._cache[name] = f
        return f


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(_FP)


def _up(a):
    return a.ctypes.data_as(_UP)


def _dp(a):
    return a.ctypes.data_as(_DP)


def max_threads() -> int:
    return int(lib().orc_max_threads())


# ---- SE(3) ------------------------------------------------------------------------------
def _vec_fn(name, n_in, n_out):
    def f(a):
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(n_in)
        o = np.zeros(n_out)
        getattr(lib(), name)(_dp(a), _dp(o))
        return o
    return f


pose_from_ypr = _vec_fn("orc_pose_from_ypr", 6, 12)
pose_to_ypr = _vec_fn("orc_pose_to_ypr", 12, 6)
se3_exp = _vec_fn("orc_se3_exp", 6, 12)
se3_log = _vec_fn("orc_se3_log", 12, 6)
so3_log = _vec_fn("orc_so3_log", 12, 3)
pose_inverse = _vec_fn("orc_pose_inverse", 12, 12)


def pose_compose(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(12)
    b = np.ascontiguousarray(b, dtype=np.float64).reshape(12)
    o = np.zeros(12)
    lib().orc_pose_compose(_dp(a), _dp(b), _dp(o))
    return o


# ---- map --------------------------------------------------------------------------------
class Map:
    def __init__(self, voxel_size=1.0, max_points_per_voxel=20, index_mode=INDEX_FLOOR, min_distance_between_points=0.0,
                 ndt_max_eigen_ratio=0.0, ndt_min_points=4, far_voxel_metric=FAR_CHEBYSHEV):
        p = _MapParams(voxel_size, max_points_per_voxel, index_mode, min_distance_between_points, ndt_max_eigen_ratio,
                       ndt_min_points, far_voxel_metric)
        self._h = lib().orc_map_create(C.byref(p))
        self.voxel_size = voxel_size

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_map_destroy(self._h)
            self._h = None

    def insert(self, xyz):
        xyz = np.asarray(xyz, dtype=np.float32)
        x, y, z = _f32(xyz[:, 0]), _f32(xyz[:, 1]), _f32(xyz[:, 2])
        lib().orc_map_insert(self._h, _fp(x), _fp(y), _fp(z), len(x))
        return self

    def insert_posed(self, xyz, T, remove_voxels_farther_than=0.0):
        """FilterMerge + insertPointCloud + far-voxel removal (orc_map_insert_posed)."""
        xyz = np.asarray(xyz, dtype=np.float32)
        x, y, z = _f32(xyz[:, 0]), _f32(xyz[:, 1]), _f32(xyz[:, 2])
        T = np.ascontiguousarray(np.asarray(T, np.float64).reshape(-1)[:12])
        lib().orc_map_insert_posed(self._h, _fp(x), _fp(y), _fp(z), len(x), _dp(T), float(remove_voxels_farther_than))
        return self

    @property
    def num_points(self):
        return int(lib().orc_map_num_points(self._h))

    @property
    def num_voxels(self):
        return int(lib().orc_map_num_voxels(self._h))

    def bbox(self):
        mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
        lib().orc_map_bbox(self._h, _fp(mn), _fp(mx))
        return mn, mx

    def dump(self):
        n, v = self.num_points, self.num_voxels
        x, y, z = (np.zeros(n, np.float32) for _ in range(3))
        src = np.zeros(n, np.uint32)
        keys = np.zeros((v, 3), np.int32)
        first, count = np.zeros(v, np.uint32), np.zeros(v, np.uint32)
        lib().orc_map_dump(self._h, _fp(x), _fp(y), _fp(z), _up(src), keys.ctypes.data_as(C.POINTER(C.c_int32)),
                           _up(first), _up(count))
        return dict(xyz=np.stack([x, y, z], 1), src_idx=src, vox_keys=keys, vox_first=first, vox_count=count)

    def dump_ndt(self):
        v = self.num_voxels
        a = [np.zeros(max(v, 1), np.float32) for _ in range(6)]
        pl = np.zeros(max(v, 1), np.uint32)
        lib().orc_map_dump_ndt(self._h, *[_fp(x) for x in a], _up(pl))
        return dict(centroid=np.stack(a[:3], 1)[:v], normal=np.stack(a[3:], 1)[:v], is_plane=pl[:v])

    def nn_single(self, q):
        pt = np.zeros(3, np.float32)
        d2 = C.c_float()
        idx = C.c_uint32()
        ok = lib().orc_map_nn_single(self._h, float(q[0]), float(q[1]), float(q[2]), _fp(pt), C.byref(d2),
                                     C.byref(idx), None, None)
        return bool(ok), pt, float(d2.value), int(idx.value)


def match_points(m: Map, local_xyz, T, threshold, threshold_angular_deg=0.0, n_threads=1):
    l = np.asarray(local_xyz, dtype=np.float32)
    n = len(l)
    lx, ly, lz = _f32(l[:, 0]), _f32(l[:, 1]), _f32(l[:, 2])
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(12)
    li, gi = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32)
    gx, gy, gz, d2 = (np.zeros(max(n, 1), np.float32) for _ in range(4))
    st = _MatchStats()
    k = lib().orc_match_points(m._h, _fp(lx), _fp(ly), _fp(lz), n, _dp(T), float(threshold),
                               float(threshold_angular_deg), _up(li), _up(gi), _fp(gx), _fp(gy), _fp(gz), _fp(d2),
                               C.byref(st), n_threads)
    return dict(local_idx=li[:k].copy(), global_idx=gi[:k].copy(), global_xyz=np.stack([gx[:k], gy[:k], gz[:k]], 1),
                d2=d2[:k].copy(), potential_pairings=int(st.potential_pairings), n_candidates=int(st.n_candidates),
                n_voxels_hit=int(st.n_voxels_hit))


def match_points_k(m: Map, local_xyz, T, threshold, k, threshold_angular_deg=0.0):
    """Matcher_Points_DistanceThreshold with pairingsPerPoint = k (orc_match_points_k)."""
    l = np.asarray(local_xyz, dtype=np.float32)
    n = len(l)
    lx, ly, lz = _f32(l[:, 0]), _f32(l[:, 1]), _f32(l[:, 2])
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(12)
    cap = max(n * int(k), 1)
    li, gi = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
    gx, gy, gz, d2 = (np.zeros(cap, np.float32) for _ in range(4))
    st = _MatchStats()
    f = lib().orc_match_points_k
    f.restype = C.c_size_t
    f.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_size_t,
                  C.POINTER(C.c_double), C.c_double, C.c_double, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                  C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p]
    np_ = f(m._h, _fp(lx), _fp(ly), _fp(lz), n, _dp(T), float(threshold), float(threshold_angular_deg), int(k), _up(li),
            _up(gi), _fp(gx), _fp(gy), _fp(gz), _fp(d2), C.byref(st))
    return dict(local_idx=li[:np_].copy(), global_idx=gi[:np_].copy(), global_xyz=np.stack([gx[:np_], gy[:np_], gz[:np_]], 1),
                d2=d2[:np_].copy(), potential_pairings=int(st.potential_pairings))


def match_pt2pl_knn(m: Map, local_xyz, T, distance_threshold, plane_eigen_threshold, search_radius, knn, minimum_plane_points):
    """Matcher_Point2Plane on a plain point map (rgbd.yaml:143-151): k nearest neighbours + PCA (orc_match_pt2pl_knn)."""
    l = np.asarray(local_xyz, dtype=np.float32).reshape(-1, 3)
    n = len(l)
    lx, ly, lz = _f32(l[:, 0]), _f32(l[:, 1]), _f32(l[:, 2])
    li = np.zeros(max(n, 1), np.uint32)
    c = [np.zeros(max(n, 1), np.float32) for _ in range(6)]
    pr = _Pt2PlKnnParams(float(distance_threshold), float(plane_eigen_threshold), float(search_radius), int(knn), int(minimum_plane_points))
    T12 = np.ascontiguousarray(T, dtype=np.float64).reshape(12)
    k = lib().orc_match_pt2pl_knn(m._h, _fp(lx), _fp(ly), _fp(lz), n, _dp(T12), C.byref(pr), _up(li), *[_fp(a) for a in c])
    return dict(local_idx=li[:k].copy(), centroid=np.stack([c[0][:k], c[1][:k], c[2][:k]], 1), normal=np.stack([c[3][:k], c[4][:k], c[5][:k]], 1))


def match_pt2pl(m: Map, local_xyz, T, distance_threshold, n_threads=1, mode=PT2PL_PLANE_DISTANCE):
    distance_threshold = (-1.0 if mode == PT2PL_CENTROID_DISTANCE else 1.0) * abs(float(distance_threshold))
    l = np.asarray(local_xyz, dtype=np.float32)
    n = len(l)
    lx, ly, lz = _f32(l[:, 0]), _f32(l[:, 1]), _f32(l[:, 2])
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(12)
    li = np.zeros(max(n, 1), np.uint32)
    a = [np.zeros(max(n, 1), np.float32) for _ in range(6)]
    k = lib().orc_match_pt2pl(m._h, _fp(lx), _fp(ly), _fp(lz), n, _dp(T), float(distance_threshold), _up(li),
                              *[_fp(x) for x in a], n_threads)
    return dict(local_idx=li[:k].copy(), centroid=np.stack(a[:3], 1)[:k].copy(), normal=np.stack(a[3:], 1)[:k].copy())


@dataclass
class GNParams:
    max_inner_iterations: int = 2
    robust_kernel: int = KERNEL_GM_C4
    robust_kernel_param: float = 1.0
    min_delta: float = 1e-7
    max_cost: float = 0.0
    weight_pt2pt: float = 1.0
    weight_pt2pl: float = 1.0

    def c(self):
        return _GNParams(self.max_inner_iterations, self.robust_kernel, self.robust_kernel_param, self.min_delta,
                         self.max_cost, self.weight_pt2pt, self.weight_pt2pl)


def _mk_prior(prior):
    if prior is None:
        return None
    mean, info = prior
    p = _Prior()
    p.mean[:] = list(np.asarray(mean, dtype=np.float64).reshape(12))
    p.info[:] = list(np.asarray(info, dtype=np.float64).reshape(36))
    return p


def _mk_pt2pt(local_xyz, global_xyz):
    l, g = np.asarray(local_xyz, np.float32).reshape(-1, 3), np.asarray(global_xyz, np.float32).reshape(-1, 3)
    arrs = [_f32(l[:, 0]), _f32(l[:, 1]), _f32(l[:, 2]), _f32(g[:, 0]), _f32(g[:, 1]), _f32(g[:, 2])]
    return _PairsPt2Pt(*[_fp(a) for a in arrs], len(l)), arrs


def _mk_pt2pl(local_xyz, centroid_xyz, normal_xyz):
    l, c, nn = (np.asarray(a, np.float32).reshape(-1, 3) for a in (local_xyz, centroid_xyz, normal_xyz))
    arrs = [_f32(a[:, i]) for a in (l, c, nn) for i in range(3)]
    return _PairsPt2Pl(*[_fp(a) for a in arrs], len(l)), arrs


def gn_solve(T, pt2pt=None, pt2pl=None, params: GNParams | None = None, prior=None, n_threads=1):
    """pt2pt = (local_xyz, global_xyz); pt2pl = (local_xyz, centroid, normal).  Returns (T_out, steps)"""
    params = params or GNParams()
    gp = params.c()
    pp, keep1 = _mk_pt2pt(*pt2pt) if pt2pt is not None else (None, None)
    pl, keep2 = _mk_pt2pl(*pt2pl) if pt2pl is not None else (None, None)
    pr = _mk_prior(prior)
    Tio = np.ascontiguousarray(T, dtype=np.float64).reshape(12).copy()
    trace = (_GNStep * max(1, params.max_inner_iterations))()
    n = lib().orc_gn_solve(C.byref(pp) if pp else None, C.byref(pl) if pl else None, C.byref(gp),
                           C.byref(pr) if pr else None, _dp(Tio), trace, n_threads)
    steps = []
    for i in range(params.max_inner_iterations):
        s = trace[i]
        steps.append(dict(H=np.array(s.H).reshape(6, 6), g=np.array(s.g), err_norm_sqr=s.err_norm_sqr,
                          delta=np.array(s.delta), T_after=np.array(s.T_after)))
    return Tio, n, steps


def covariance(T, pt2pt=None, pt2pl=None, findif_xyz=1e-7, findif_ang=1e-7):
    pp, keep1 = _mk_pt2pt(*pt2pt) if pt2pt is not None else (None, None)
    pl, keep2 = _mk_pt2pl(*pt2pl) if pt2pl is not None else (None, None)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(12)
    cov, ata = np.zeros(36), np.zeros(36)
    lib().orc_covariance(C.byref(pp) if pp else None, C.byref(pl) if pl else None, _dp(T), findif_xyz, findif_ang,
                         _dp(cov), _dp(ata))
    return cov.reshape(6, 6), ata.reshape(6, 6)


@dataclass
class ICPParams:
    max_iterations: int = 300
    min_abs_step_trans: float = 1e-4
    min_abs_step_rot: float = 5e-5
    disable_stall_test: bool = False
    threshold: object = None  # array [max_iterations]
    threshold_angular_deg: float = 0.0
    pt2pl_threshold: object = None  # None = no Matcher_Point2Plane, else array [max_iterations]
    pt2pl_mode: int = 0             # PT2PL_PLANE_DISTANCE | PT2PL_CENTROID_DISTANCE
    kernel_param: object = None  # array [max_iterations]
    gn: GNParams = field(default_factory=GNParams)
    hook_enabled: bool = False
    hook_min_trans: float = 0.15
    hook_min_rot: float = np.deg2rad(0.75)
    hook_checkpoint: object = None
    compute_covariance: bool = True
    cov_findif_xyz: float = 1e-7
    cov_findif_ang: float = 1e-7
    pt2pt_skip_plane_paired: bool = False  # U12: points paired by Matcher_Point2Plane are skipped by the point matcher


def icp_align(m: Map, local_xyz, T_guess, p: ICPParams, prior=None, n_threads=1, want_pairs=False):
    l = np.asarray(local_xyz, dtype=np.float32).reshape(-1, 3)
    n = len(l)
    lx, ly, lz = _f32(l[:, 0]), _f32(l[:, 1]), _f32(l[:, 2])
    thr = np.ascontiguousarray(np.broadcast_to(np.asarray(p.threshold, np.float64), (p.max_iterations,)))
    kp = np.ascontiguousarray(np.broadcast_to(np.asarray(p.kernel_param, np.float64), (p.max_iterations,)))
    cp = _ICPParams()
    cp.max_iterations = p.max_iterations
    cp.min_abs_step_trans = p.min_abs_step_trans
    cp.min_abs_step_rot = p.min_abs_step_rot
    cp.disable_stall_test = int(p.disable_stall_test)
    cp.pt2pt_skip_plane_paired = int(p.pt2pt_skip_plane_paired)
    cp.threshold = _dp(thr)
    cp.threshold_angular_deg = p.threshold_angular_deg
    cp.kernel_param = _dp(kp)
    plt = None
    if p.pt2pl_threshold is not None:
        plt = np.ascontiguousarray(np.broadcast_to(np.asarray(p.pt2pl_threshold, np.float64), (p.max_iterations,)))
        if p.pt2pl_mode == PT2PL_CENTROID_DISTANCE:
            plt = -np.abs(plt)
        cp.pt2pl_threshold = _dp(plt)
    cp.gn = p.gn.c()
    cp.hook_enabled = int(p.hook_enabled)
    cp.hook_min_trans = p.hook_min_trans
    cp.hook_min_rot = p.hook_min_rot
    chk = np.asarray(p.hook_checkpoint if p.hook_checkpoint is not None else T_guess, np.float64).reshape(12)
    cp.hook_checkpoint[:] = list(chk)
    cp.compute_covariance = int(p.compute_covariance)
    cp.cov_findif_xyz = p.cov_findif_xyz
    cp.cov_findif_ang = p.cov_findif_ang
    T0 = np.ascontiguousarray(T_guess, dtype=np.float64).reshape(12)
    res = _ICPResult()
    trace = (_ICPIter * max(1, p.max_iterations))()
    pr = _mk_prior(prior)
    po = None
    if want_pairs:
        li, gi = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32)
        gx, gy, gz, d2 = (np.zeros(max(n, 1), np.float32) for _ in range(4))
        po = _PairsOut(_up(li), _up(gi), _fp(gx), _fp(gy), _fp(gz), _fp(d2))
    rc = lib().orc_icp_align(m._h, _fp(lx), _fp(ly), _fp(lz), n, _dp(T0), C.byref(cp),
                             C.byref(pr) if pr else None, C.byref(res), trace, C.byref(po) if po else None,
                             n_threads)
    assert rc == 0
    n_tr = min(p.max_iterations, res.n_iterations + 1) if p.max_iterations else 0
    out = dict(T=np.array(res.T), cov=np.array(res.cov).reshape(6, 6), quality=res.quality,
               n_iterations=int(res.n_iterations), termination_reason=int(res.termination_reason),
               n_final_pairs=int(res.n_final_pairs), potential_pairings=int(res.potential_pairings),
               n_candidates_total=int(res.n_candidates_total), n_final_pairs_pt2pl=int(res.n_final_pairs_pt2pl),
               trace=[dict(T=np.array(trace[i].T), n_pairs=int(trace[i].n_pairs), threshold=trace[i].threshold,
                           kernel_param=trace[i].kernel_param, delta_trans=trace[i].delta_trans,
                           delta_rot=trace[i].delta_rot) for i in range(n_tr)])
    if want_pairs:
        k = out["n_final_pairs"] - out["n_final_pairs_pt2pl"]
        out["pairs"] = dict(local_idx=li[:k].copy(), global_idx=gi[:k].copy(),
                            global_xyz=np.stack([gx[:k], gy[:k], gz[:k]], 1), d2=d2[:k].copy())
    return out


# ---- scan pre-processing ------------------------------------------------------------------
TS_NONE, TS_MIDDLE_IS_ZERO, TS_EARLIEST_IS_ZERO = 0, 1, 2


def _xyz_cols(xyz):
    xyz = np.asarray(xyz, dtype=np.float32)
    return _f32(xyz[:, 0]), _f32(xyz[:, 1]), _f32(xyz[:, 2])


def adjust_timestamps(t, method=TS_MIDDLE_IS_ZERO, time_offset=0.0):
    t = _f32(np.array(t, dtype=np.float32, copy=True))
    lib().orc_adjust_timestamps(_fp(t), len(t), int(method), float(time_offset))
    return t


def decimate_first_point(xyz, resolution, min_points_to_filter=0, index_mode=INDEX_FLOOR):
    x, y, z = _xyz_cols(xyz)
    out = np.zeros(max(len(x), 1), np.uint32)
    k = lib().orc_decimate_first_point(_fp(x), _fp(y), _fp(z), len(x), float(resolution), int(min_points_to_filter),
                                       int(index_mode), _up(out))
    return out[:k].copy()


def filter_by_range(xyz, range_min, range_max, center=(0.0, 0.0, 0.0)):
    x, y, z = _xyz_cols(xyz)
    out = np.zeros(max(len(x), 1), np.uint32)
    c = _f32(np.asarray(center, np.float32))
    k = lib().orc_filter_by_range(_fp(x), _fp(y), _fp(z), len(x), float(range_min), float(range_max), _fp(c), _up(out))
    return out[:k].copy()


def filter_bbox(xyz, bb_min, bb_max, keep_inside=False):
    x, y, z = _xyz_cols(xyz)
    out = np.zeros(max(len(x), 1), np.uint32)
    mn, mx = _f32(np.asarray(bb_min, np.float32)), _f32(np.asarray(bb_max, np.float32))
    k = lib().orc_filter_bbox(_fp(x), _fp(y), _fp(z), len(x), _fp(mn), _fp(mx), int(bool(keep_inside)), _up(out))
    return out[:k].copy()


def deskew(xyz, t, twist):
    x, y, z = _xyz_cols(xyz)
    t = _f32(t)
    tw = np.ascontiguousarray(twist, dtype=np.float64)
    ox, oy, oz = (np.zeros(len(x), np.float32) for _ in range(3))
    lib().orc_deskew(_fp(x), _fp(y), _fp(z), _fp(t), len(x), _dp(tw), _fp(ox), _fp(oy), _fp(oz))
    return np.stack([ox, oy, oz], 1)


def decimate_closest_to_average(xyz, resolution, min_points_to_filter=0, index_mode=INDEX_FLOOR):
    """FilterDecimateVoxels(ClosestToAverage) [U] -> surviving input indices, ascending."""
    x, y, z = _xyz_cols(xyz)
    out = np.zeros(max(len(x), 1), np.uint32)
    k = lib().orc_decimate_closest_to_average(_fp(x), _fp(y), _fp(z), len(x), float(resolution), int(min_points_to_filter),
                                              int(index_mode), _up(out))
    return out[:k].copy()


def preprocess(xyz, decim_map_resolution, decim_icp_resolution, min_points_to_filter=2000, index_mode=INDEX_FLOOR,
               range_min=0.0, range_max=0.0, range_center=(0.0, 0.0, 0.0), bbox_mode=0, bbox_min=(0, 0, 0),
               bbox_max=(0, 0, 0), decim_map_method=0, decim_icp_method=0):
    """1st-pass filter chain of lidar3d-default.yaml:278-319 -> (idx_map, idx_icp), indices into the raw scan.
    decim_*_method: DECIMATE_FIRST_POINT | DECIMATE_CLOSEST_TO_AVERAGE (decimate_method of the two FilterDecimateVoxels)."""
    x, y, z = _xyz_cols(xyz)
    p = _PreprocessParams(float(decim_map_resolution), float(decim_icp_resolution), int(min_points_to_filter),
                          int(index_mode), float(range_min), float(range_max), (C.c_float * 3)(*map(float, range_center)),
                          int(bbox_mode), (C.c_float * 3)(*map(float, bbox_min)), (C.c_float * 3)(*map(float, bbox_max)),
                          int(decim_map_method), int(decim_icp_method))
    im, ii = np.zeros(max(len(x), 1), np.uint32), np.zeros(max(len(x), 1), np.uint32)
    nm, ni = C.c_size_t(0), C.c_size_t(0)
    lib().orc_preprocess(_fp(x), _fp(y), _fp(z), len(x), C.byref(p), _up(im), C.byref(nm), _up(ii), C.byref(ni))
    return im[:nm.value].copy(), ii[:ni.value].copy()
