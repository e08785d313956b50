"""ctypes binding of libmolahip.so -- the C ABI declared in include/molahip.h.

This is plumbing for tests and bench.py: every call goes straight through the C ABI, i.e. through
exactly the entry points an mp2p_icp plugin adapter would bind.  There is no
Python or CPU fallback: if the shared library is missing or no HIP device is present the calls
raise (MolahipError / OSError).
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmolahip.so")
if os.environ.get("MOLAHIP_LIB_PATH"):  # development: an A/B build of the same sources (tools/build_variants.sh)
    LIB_PATH = os.environ["MOLAHIP_LIB_PATH"]

MH_OK = 0
MH_WARN_PREVIOUS_OUT_OF_RANGE = 64  # not a failure: see mh_map_insert in include/molahip.h
MEM_HOST, MEM_DEVICE, MEM_HOST_PINNED = 0, 1, 2
INDEX_FLOOR, INDEX_TRUNC = 0, 1
FAR_CHEBYSHEV, FAR_L1, FAR_L2 = 0, 1, 2              # mh_map_params::far_voxel_metric
PT2PL_PLANE_DISTANCE, PT2PL_CENTROID_DISTANCE = 0, 1  # mh_icp_params::pt2pl_mode
KERNEL_NONE, KERNEL_GM_C4, KERNEL_GM_KISS, KERNEL_GM_BARRON, KERNEL_CAUCHY, KERNEL_GM_C2 = range(6)
TERM_NAMES = ["Undefined", "NoPairings", "SolverError", "MaxIterations", "Stalled",
              "QualityCheckpointFailed", "HookRequest"]
NO_MATCH = 0xFFFFFFFF

_FP = C.POINTER(C.c_float)
_UP = C.POINTER(C.c_uint32)
_DP = C.POINTER(C.c_double)


class MolahipError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"libmolahip status {status}: {msg}")
        self.status = status


class MapParams(C.Structure):
    _fields_ = [("voxel_size", C.c_float), ("max_points_per_voxel", C.c_uint32), ("index_mode", C.c_uint32),
                ("min_distance_between_points", C.c_float), ("ndt_max_eigen_ratio", C.c_float),
                ("ndt_min_points", C.c_uint32), ("far_voxel_metric", C.c_uint32)]


class MapInfo(C.Structure):
    _fields_ = [("n_points", C.c_uint64), ("n_offered", C.c_uint64), ("n_voxels", C.c_uint64),
                ("table_size", C.c_uint64), ("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3),
                ("voxel_size", C.c_float), ("max_points_per_voxel", C.c_uint32), ("n_planes", C.c_uint64),
                ("deferred_status", C.c_uint32), ("reserved_", C.c_uint32)]


class PairsOut(C.Structure):
    _fields_ = [("local_idx", _UP), ("global_idx", _UP), ("gx", _FP), ("gy", _FP), ("gz", _FP), ("d2", _FP)]


class Pt2PlKnnParams(C.Structure):
    _fields_ = [("distance_threshold", C.c_double), ("plane_eigen_threshold", C.c_double), ("search_radius", C.c_double),
                ("knn", C.c_uint32), ("minimum_plane_points", C.c_uint32)]


class PairsPlOut(C.Structure):
    _fields_ = [("local_idx", _UP), ("cx", _FP), ("cy", _FP), ("cz", _FP), ("nx", _FP), ("ny", _FP), ("nz", _FP)]


class MatchInfo(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("potential_pairings", C.c_uint64)]


class PairsPt2Pt(C.Structure):
    _fields_ = [("lx", _FP), ("ly", _FP), ("lz", _FP), ("gx", _FP), ("gy", _FP), ("gz", _FP), ("n", C.c_size_t)]


class PairsPt2Pl(C.Structure):
    _fields_ = [("lx", _FP), ("ly", _FP), ("lz", _FP), ("cx", _FP), ("cy", _FP), ("cz", _FP),
                ("nx", _FP), ("ny", _FP), ("nz", _FP), ("n", C.c_size_t)]


class Prior(C.Structure):
    _fields_ = [("mean", C.c_double * 12), ("info", C.c_double * 36)]


class GNParamsC(C.Structure):
    _fields_ = [("max_inner_iterations", C.c_uint32), ("robust_kernel", C.c_uint32),
                ("robust_kernel_param", C.c_double), ("min_delta", C.c_double), ("max_cost", C.c_double),
                ("weight_pt2pt", C.c_double), ("weight_pt2pl", C.c_double)]


class GNStep(C.Structure):
    _fields_ = [("H", C.c_double * 36), ("g", C.c_double * 6), ("err_norm_sqr", C.c_double),
                ("delta", C.c_double * 6), ("T_after", C.c_double * 12)]


class ICPParamsC(C.Structure):
    _fields_ = [("max_iterations", C.c_uint32), ("min_abs_step_trans", C.c_double), ("min_abs_step_rot", C.c_double),
                ("disable_stall_test", C.c_uint32), ("threshold", _DP), ("kernel_param", _DP),
                ("threshold_angular_deg", C.c_double), ("pt2pl_threshold", _DP), ("gn", GNParamsC), ("hook_enabled", C.c_uint32),
                ("hook_min_trans", C.c_double), ("hook_min_rot", C.c_double), ("hook_checkpoint", C.c_double * 12),
                ("compute_covariance", C.c_uint32), ("cov_findif_xyz", C.c_double), ("cov_findif_ang", C.c_double),
                ("poll_every", C.c_uint32), ("expected_iterations", C.c_uint32), ("pt2pl_mode", C.c_uint32),
                ("matched_points", C.c_uint32),
                ("profile", C.c_uint32)]


class ICPIter(C.Structure):
    _fields_ = [("T", C.c_double * 12), ("n_pairs", C.c_uint32), ("threshold", C.c_double),
                ("kernel_param", C.c_double), ("delta_trans", C.c_double), ("delta_rot", C.c_double)]


class ICPResult(C.Structure):
    _fields_ = [("T", C.c_double * 12), ("cov", C.c_double * 36), ("quality", C.c_double),
                ("n_iterations", C.c_uint32), ("termination_reason", C.c_uint32), ("n_final_pairs", C.c_uint32),
                ("potential_pairings", C.c_uint64), ("n_match_launches", C.c_uint32),
                ("match_kernel_ms", C.c_double), ("total_ms", C.c_double), ("n_final_pairs_pt2pl", C.c_uint32),
                ("n_host_polls", C.c_uint32), ("n_enqueued_iterations", C.c_uint32)]


class PreprocessParams(C.Structure):
    _fields_ = [("decim_map_resolution", C.c_float), ("decim_icp_resolution", C.c_float),
                ("min_points_to_filter", C.c_uint32), ("index_mode", C.c_int32), ("range_min", C.c_float),
                ("range_max", C.c_float), ("range_center", C.c_float * 3), ("bbox_mode", C.c_int32),
                ("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3), ("timestamp_method", C.c_int32),
                ("time_offset", C.c_float), ("decim_map_method", C.c_int32), ("decim_icp_method", C.c_int32)]


TS_NONE, TS_MIDDLE_IS_ZERO, TS_EARLIEST_IS_ZERO = 0, 1, 2
DECIMATE_FIRST_POINT, DECIMATE_CLOSEST_TO_AVERAGE = 0, 1
BBOX_OFF, BBOX_KEEP_OUTSIDE, BBOX_KEEP_INSIDE = 0, 1, 2


# every entry point include/molahip.h declares, with its ctypes signature
_SIGNATURES = {
    "mh_version": (C.c_int32, [_UP, _UP, _UP]),
    "mh_last_error_string": (C.c_char_p, []),
    "mh_status_string": (C.c_char_p, [C.c_int32]),
    "mh_device_count": (C.c_int32, [C.POINTER(C.c_int32)]),
    "mh_debug_fail_allocations": (C.c_int32, [C.c_int32, C.c_int32]),
    "mh_debug_loop_stats": (None, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "mh_debug_dev_variants": (C.c_int32, []),
    "mh_abi_version": (C.c_uint32, []),
    "mh_icp_align_prefers_solo": (C.c_int32, [C.c_void_p, C.POINTER(ICPParamsC), C.c_uint32, C.POINTER(C.c_int32)]),
    "mh_ctx_create": (C.c_int32, [C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "mh_ctx_destroy": (C.c_int32, [C.c_void_p]),
    "mh_ctx_synchronize": (C.c_int32, [C.c_void_p]),
    "mh_ctx_stream": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mh_ctx_memory_info": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "mh_map_create": (C.c_int32, [C.c_void_p, C.POINTER(MapParams), C.POINTER(C.c_void_p)]),
    "mh_map_destroy": (C.c_int32, [C.c_void_p]),
    "mh_map_build": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32]),
    "mh_map_get_info": (C.c_int32, [C.c_void_p, C.POINTER(MapInfo)]),
    "mh_map_download": (C.c_int32, [C.c_void_p, _FP, _FP, _FP, _UP, C.POINTER(C.c_int32), _UP, _UP]),
    "mh_map_download_ndt": (C.c_int32, [C.c_void_p, _FP, _FP, _FP, _FP, _FP, _FP, _UP]),
    "mh_scan_create": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32,
                                   C.POINTER(C.c_void_p)]),
    "mh_scan_update": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32]),
    "mh_scan_update_aos": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                       C.c_int64, C.c_int32]),
    "mh_scan_prepare": (C.c_int32, [C.c_void_p, C.c_float]),
    "mh_scan_destroy": (C.c_int32, [C.c_void_p]),
    "mh_scan_size": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "mh_map_insert": (C.c_int32, [C.c_void_p, C.c_void_p, _DP, C.c_float]),
    "mh_scan_set_timestamps": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32]),
    "mh_scan_preprocess": (C.c_int32, [C.c_void_p, C.POINTER(PreprocessParams), C.c_void_p, C.c_void_p]),
    "mh_scan_preprocess_batch": (C.c_int32, [C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(PreprocessParams), C.c_size_t,
                                             C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "mh_scan_deskew": (C.c_int32, [C.c_void_p, _DP, C.c_void_p]),
    "mh_scan_deskew_pair": (C.c_int32, [C.c_void_p, C.c_void_p, _DP, C.c_void_p, C.c_void_p, _FP, _FP, C.POINTER(C.c_uint64)]),
    "mh_host_alloc_pinned": (C.c_int32, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "mh_host_free_pinned": (C.c_int32, [C.c_void_p]),
    "mh_scan_download": (C.c_int32, [C.c_void_p, _FP, _FP, _FP, _FP, _UP]),
    "mh_scan_bbox": (C.c_int32, [C.c_void_p, _FP, _FP, C.POINTER(C.c_uint64)]),
    "mh_nn_search": (C.c_int32, [C.c_void_p, C.c_void_p, _DP, C.c_double, C.c_double, C.POINTER(PairsOut), C.c_int32,
                                 C.POINTER(MatchInfo)]),
    "mh_nn_search_k": (C.c_int32, [C.c_void_p, C.c_void_p, _DP, C.c_double, C.c_double, C.c_uint32, C.POINTER(PairsOut),
                                   C.c_int32, C.POINTER(MatchInfo)]),
    "mh_nn_search_dense": (C.c_int32, [C.c_void_p, C.c_void_p, _DP, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int32]),
    "mh_nn_search_pt2pl": (C.c_int32, [C.c_void_p, C.c_void_p, _DP, C.c_double, C.c_uint32, C.POINTER(PairsPlOut), C.c_int32,
                                       C.POINTER(MatchInfo)]),
    "mh_nn_search_pt2pl_knn": (C.c_int32, [C.c_void_p, C.c_void_p, _DP, C.POINTER(Pt2PlKnnParams), C.POINTER(PairsPlOut), C.c_int32,
                                           C.POINTER(MatchInfo)]),
    "mh_icp_get_pt2pl_pairs": (C.c_int32, [C.c_void_p, C.POINTER(PairsPlOut), C.c_int32, C.POINTER(C.c_uint64)]),
    "mh_gn_solve": (C.c_int32, [C.c_void_p, C.POINTER(PairsPt2Pt), C.POINTER(PairsPt2Pl), C.c_int32,
                                C.POINTER(GNParamsC), C.POINTER(Prior), _DP, C.POINTER(C.c_int32),
                                C.POINTER(C.c_int32), C.POINTER(GNStep)]),
    "mh_covariance": (C.c_int32, [C.c_void_p, C.POINTER(PairsPt2Pt), C.POINTER(PairsPt2Pl), C.c_int32, _DP, C.c_double,
                                  C.c_double, _DP]),
    "mh_icp_align": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(ICPParamsC), _DP, C.POINTER(Prior),
                                 C.POINTER(ICPResult), C.POINTER(ICPIter), C.POINTER(PairsOut), C.c_int32]),
    "mh_icp_align_batch": (C.c_int32, [C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(ICPParamsC),
                                       C.c_int32, _DP, C.POINTER(C.POINTER(Prior)), C.POINTER(ICPResult), C.c_void_p,
                                       C.c_int32]),
    "mh_pairs_block_bytes": (C.c_size_t, [C.c_size_t]),
}

_lib = None


def lib():
    """Load libmolahip.so (built by __graft_entry__.build() / csrc/Makefile).  Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)")
        # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64; if torch is going
        # to be used in this process (bench.py, device-pointer interop) it must be loaded FIRST so that
        # libmolahip binds to the same runtime -- otherwise the second runtime finds "no HIP GPUs".
        try:
            import torch  # noqa: F401  (plumbing only; libmolahip itself does not depend on torch)
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        want = _header_abi_version()
        if want is not None and int(L.mh_abi_version()) != want:
            raise OSError(f"{LIB_PATH} speaks ABI version {int(L.mh_abi_version())}, include/molahip.h says {want}: rebuild "
                          "(the parameter structs of this binding follow the header)")
        _lib = L
    return _lib


def _header_abi_version():
    """MH_ABI_VERSION of include/molahip.h beside this package (None when the header is not there: an installed copy)."""
    import re
    h = os.path.join(os.path.dirname(_HERE), "include", "molahip.h")
    try:
        m = re.search(r"^#define\s+MH_ABI_VERSION\s+(\d+)", open(h).read(), re.M)
    except OSError:
        return None
    return int(m.group(1)) if m else None


def _chk(status):
    if status != MH_OK:
        raise MolahipError(status, lib().mh_last_error_string().decode(errors="replace"))


def fail_allocations(first_attempts: int, retries: int = 0):
    """Fault injection (mh_debug_fail_allocations): the next device allocations of the library's buffers fail."""
    _chk(lib().mh_debug_fail_allocations(int(first_attempts), int(retries)))


def loop_stats():
    """(one-launch loops started, loops abandoned for the launch-by-launch chain) so far in this process (mh_debug_loop_stats)."""
    a, b = C.c_uint64(0), C.c_uint64(0)
    lib().mh_debug_loop_stats(C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


def dev_variants() -> bool:
    """True when the loaded library is the development build (tools/build_variants.sh): MH_MATCH=t|w|o are selectable."""
    return bool(lib().mh_debug_dev_variants())


def device_count() -> int:
    n = C.c_int32(0)
    st = lib().mh_device_count(C.byref(n))
    return int(n.value) if st == MH_OK else 0


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _soa(xyz):
    xyz = np.asarray(xyz, dtype=np.float32).reshape(-1, 3)
    return _f32(xyz[:, 0]), _f32(xyz[:, 1]), _f32(xyz[:, 2])


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def _T12(T):
    T = np.ascontiguousarray(T, dtype=np.float64)
    if T.size == 16:
        T = T.reshape(4, 4)[:3]
    return np.ascontiguousarray(T.reshape(12))


class Context:
    """One HIP device + one stream (mh_ctx)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        """stream: a hipStream_t (e.g. torch's) to run on, or None: the context creates and owns a non-blocking stream"""
        self._h = C.c_void_p()
        _chk(lib().mh_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(self._h)))
        self.device = device
        self._children = weakref.WeakSet()  # maps/scans must be destroyed before their context (C-ABI rule)

    def synchronize(self):
        _chk(lib().mh_ctx_synchronize(self._h))

    def memory_info(self):
        """(free, total) bytes of device memory."""
        f, t = C.c_uint64(), C.c_uint64()
        _chk(lib().mh_ctx_memory_info(self._h, C.byref(f), C.byref(t)))
        return int(f.value), int(t.value)

    @property
    def stream(self) -> int:
        s = C.c_void_p()
        _chk(lib().mh_ctx_stream(self._h, C.byref(s)))
        return s.value or 0

    def close(self):
        if self._h:
            for child in list(self._children):
                child.close()
            lib().mh_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Map:
    """Device-resident voxel-hashed local map (stands in for mola::HashedVoxelPointCloud)."""

    def __init__(self, ctx: Context, voxel_size=1.0, max_points_per_voxel=20, index_mode=INDEX_FLOOR,
                 min_distance_between_points=0.0, ndt_max_eigen_ratio=0.0, ndt_min_points=4, far_voxel_metric=FAR_CHEBYSHEV):
        self.ctx = ctx
        self._h = C.c_void_p()
        p = MapParams(voxel_size, max_points_per_voxel, index_mode, min_distance_between_points, ndt_max_eigen_ratio,
                      ndt_min_points, far_voxel_metric)
        _chk(lib().mh_map_create(ctx._h, C.byref(p), C.byref(self._h)))
        ctx._children.add(self)

    def build(self, xyz):
        x, y, z = _soa(xyz)
        _chk(lib().mh_map_build(self._h, _vp(x), _vp(y), _vp(z), len(x), MEM_HOST))
        return self

    def build_device(self, x_ptr, y_ptr, z_ptr, n):
        _chk(lib().mh_map_build(self._h, C.c_void_p(x_ptr), C.c_void_p(y_ptr), C.c_void_p(z_ptr), n, MEM_DEVICE))
        return self

    def insert(self, scan: "Scan", T, remove_voxels_farther_than=0.0):
        """Key-frame update on the device: FilterMerge + insertPointCloud + far-voxel removal (mh_map_insert)."""
        T = _T12(T)
        st = lib().mh_map_insert(self._h, scan._h, T.ctypes.data_as(_DP), float(remove_voxels_farther_than))
        if st == MH_WARN_PREVIOUS_OUT_OF_RANGE:  # inserted; the PREVIOUS update left out-of-range points out (not a failure)
            import warnings
            warnings.warn(lib().mh_last_error_string().decode(errors="replace"), RuntimeWarning, stacklevel=2)
            return self
        _chk(st)
        return self

    def info(self) -> MapInfo:
        i = MapInfo()
        _chk(lib().mh_map_get_info(self._h, C.byref(i)))
        return i

    def download(self):
        i = self.info()
        n, v = int(i.n_points), int(i.n_voxels)
        x, y, z = (np.zeros(max(n, 1), np.float32) for _ in range(3))
        src = np.zeros(max(n, 1), np.uint32)
        keys = np.zeros((max(v, 1), 3), np.int32)
        first, count = np.zeros(max(v, 1), np.uint32), np.zeros(max(v, 1), np.uint32)
        _chk(lib().mh_map_download(self._h, x.ctypes.data_as(_FP), y.ctypes.data_as(_FP), z.ctypes.data_as(_FP),
                                   src.ctypes.data_as(_UP), keys.ctypes.data_as(C.POINTER(C.c_int32)),
                                   first.ctypes.data_as(_UP), count.ctypes.data_as(_UP)))
        return dict(xyz=np.stack([x[:n], y[:n], z[:n]], 1), src_idx=src[:n], vox_keys=keys[:v], vox_first=first[:v],
                    vox_count=count[:v])

    def download_ndt(self):
        v = int(self.info().n_voxels)
        a = [np.zeros(max(v, 1), np.float32) for _ in range(6)]
        pl = np.zeros(max(v, 1), np.uint32)
        _chk(lib().mh_map_download_ndt(self._h, *[x.ctypes.data_as(_FP) for x in a], pl.ctypes.data_as(_UP)))
        return dict(centroid=np.stack(a[:3], 1)[:v], normal=np.stack(a[3:], 1)[:v], is_plane=pl[:v])

    def close(self):
        if self._h:
            if self.ctx._h:  # (see Scan.close)
                lib().mh_map_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scan:
    """Device-resident local point layer (the `decimated_for_icp` layer handed to align())."""

    def __init__(self, ctx: Context, xyz=None, device_ptrs=None, n=None):
        self.ctx = ctx
        self._h = C.c_void_p()
        if device_ptrs is not None:
            px, py, pz = device_ptrs
            _chk(lib().mh_scan_create(ctx._h, C.c_void_p(px), C.c_void_p(py), C.c_void_p(pz), n, MEM_DEVICE,
                                      C.byref(self._h)))
        else:
            x, y, z = _soa(xyz if xyz is not None else np.zeros((0, 3), np.float32))
            _chk(lib().mh_scan_create(ctx._h, _vp(x), _vp(y), _vp(z), len(x), MEM_HOST, C.byref(self._h)))
        ctx._children.add(self)

    @property
    def n(self) -> int:
        return len(self)

    def set_timestamps(self, t):
        t = _f32(t)
        _chk(lib().mh_scan_set_timestamps(self._h, _vp(t), len(t), MEM_HOST))
        return self

    def preprocess(self, params: "PreprocessParams", out_map: "Scan", out_icp: "Scan | None" = None):
        """1st-pass observation filters on the device (mh_scan_preprocess): self = raw scan."""
        _chk(lib().mh_scan_preprocess(self._h, C.byref(params), out_map._h, out_icp._h if out_icp is not None else None))
        return out_map, out_icp

    def deskew_pair(self, small: "Scan", twist, out: "Scan", out_small: "Scan"):
        """mh_scan_deskew_pair: self = the large layer; returns (bb_min, bb_max, n_finite) of the de-skewed small layer"""
        tw = None if twist is None else np.ascontiguousarray(twist, dtype=np.float64)
        mn, mx, nf = np.zeros(3, np.float32), np.zeros(3, np.float32), C.c_uint64(0)
        _chk(lib().mh_scan_deskew_pair(self._h, small._h, tw.ctypes.data_as(_DP) if tw is not None else None, out._h, out_small._h,
                                       mn.ctypes.data_as(_FP), mx.ctypes.data_as(_FP), C.byref(nf)))
        return mn, mx, nf.value

    def deskew(self, twist, out: "Scan"):
        tw = None if twist is None else np.ascontiguousarray(twist, dtype=np.float64)
        _chk(lib().mh_scan_deskew(self._h, tw.ctypes.data_as(_DP) if tw is not None else None, out._h))
        return out

    def bbox(self):
        mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
        k = C.c_uint64()
        _chk(lib().mh_scan_bbox(self._h, mn.ctypes.data_as(_FP), mx.ctypes.data_as(_FP), C.byref(k)))
        return mn, mx, int(k.value)

    def download(self):
        n = len(self)
        x, y, z, t = (np.zeros(max(n, 1), np.float32) for _ in range(4))
        src = np.zeros(max(n, 1), np.uint32)
        _chk(lib().mh_scan_download(self._h, *[a.ctypes.data_as(_FP) for a in (x, y, z, t)], src.ctypes.data_as(_UP)))
        return dict(xyz=np.stack([x[:n], y[:n], z[:n]], 1), t=t[:n], src_idx=src[:n])

    @classmethod
    def from_torch(cls, ctx: Context, x, y, z):
        """x,y,z: contiguous float32 torch tensors on ctx's device (read in place, then copied)."""
        import torch  # plumbing only
        for t in (x, y, z):
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        torch.cuda.current_stream(x.device).synchronize()
        return cls(ctx, device_ptrs=(x.data_ptr(), y.data_ptr(), z.data_ptr()), n=x.numel())

    def update(self, xyz):
        x, y, z = _soa(xyz)
        _chk(lib().mh_scan_update(self._h, _vp(x), _vp(y), _vp(z), len(x), MEM_HOST))

    def prepare(self, voxel_size: float):
        """Queue the build of the tile matcher's search order behind whatever was uploaded last (asynchronous)."""
        _chk(lib().mh_scan_prepare(self._h, float(voxel_size)))
        return self

    def update_pinned(self, x_ptr: int, y_ptr: int, z_ptr: int, n: int):
        """Asynchronous upload from page-locked host arrays (MH_MEM_HOST_PINNED): returns at once; the caller keeps the
        arrays alive and unchanged until the context's stream has passed the copies."""
        _chk(lib().mh_scan_update(self._h, C.c_void_p(x_ptr), C.c_void_p(y_ptr), C.c_void_p(z_ptr), n, MEM_HOST_PINNED))

    def update_interleaved_pinned(self, ptr: int, n: int, point_step: int, off_x=0, off_y=4, off_z=8, off_t=-1):
        """Asynchronous upload of n interleaved records from page-locked host memory (MH_MEM_HOST_PINNED): ONE copy of the
        raw bytes + the de-interleave kernel, queued on the context's stream; returns at once."""
        _chk(lib().mh_scan_update_aos(self._h, C.c_void_p(ptr), n, point_step, off_x, off_y, off_z, off_t, MEM_HOST_PINNED))

    def update_interleaved(self, records, off_x=0, off_y=4, off_z=8, off_t=-1):
        """records: C-contiguous float32 [n,k] rows (KITTI .bin: k=4) -- one copy, de-interleaved on the device."""
        if hasattr(records, "data_ptr"):  # a contiguous float32 torch tensor on the context's device: read in place
            import torch  # plumbing only
            assert records.is_cuda and records.dtype == torch.float32 and records.is_contiguous() and records.dim() == 2
            torch.cuda.current_stream(records.device).synchronize()
            _chk(lib().mh_scan_update_aos(self._h, C.c_void_p(records.data_ptr()), records.shape[0], records.shape[1] * 4,
                                          off_x, off_y, off_z, off_t, MEM_DEVICE))
            return
        a = np.ascontiguousarray(records, dtype=np.float32)
        assert a.ndim == 2
        _chk(lib().mh_scan_update_aos(self._h, _vp(a), a.shape[0], a.shape[1] * 4, off_x, off_y, off_z, off_t, MEM_HOST))

    def __len__(self):
        n = C.c_uint64()
        _chk(lib().mh_scan_size(self._h, C.byref(n)))
        return int(n.value)

    def close(self):
        if self._h:
            # (garbage collection of a reference cycle may finalise the context first -- its weak set of children is
            # cleared before any finaliser runs: a scan whose context is gone must not touch it any more)
            if self.ctx._h:
                lib().mh_scan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def nn_search(m: Map, s: Scan, T, threshold, threshold_angular_deg=0.0):
    n = max(s.n, 1)
    li, gi = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    gx, gy, gz, d2 = (np.zeros(n, np.float32) for _ in range(4))
    out = PairsOut(li.ctypes.data_as(_UP), gi.ctypes.data_as(_UP), gx.ctypes.data_as(_FP), gy.ctypes.data_as(_FP),
                   gz.ctypes.data_as(_FP), d2.ctypes.data_as(_FP))
    info = MatchInfo()
    T = _T12(T)
    _chk(lib().mh_nn_search(m._h, s._h, T.ctypes.data_as(_DP), float(threshold), float(threshold_angular_deg),
                            C.byref(out), MEM_HOST, C.byref(info)))
    k = int(info.n_pairs)
    return dict(local_idx=li[:k].copy(), global_idx=gi[:k].copy(), global_xyz=np.stack([gx[:k], gy[:k], gz[:k]], 1),
                d2=d2[:k].copy(), potential_pairings=int(info.potential_pairings))


def nn_search_k(m: Map, s: Scan, T, threshold, k, threshold_angular_deg=0.0):
    """Matcher_Points_DistanceThreshold with pairingsPerPoint = k (mh_nn_search_k)."""
    n = max(s.n * int(k), 1)
    li, gi = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    gx, gy, gz, d2 = (np.zeros(n, np.float32) for _ in range(4))
    out = PairsOut(li.ctypes.data_as(_UP), gi.ctypes.data_as(_UP), gx.ctypes.data_as(_FP), gy.ctypes.data_as(_FP),
                   gz.ctypes.data_as(_FP), d2.ctypes.data_as(_FP))
    info = MatchInfo()
    T = _T12(T)
    _chk(lib().mh_nn_search_k(m._h, s._h, T.ctypes.data_as(_DP), float(threshold), float(threshold_angular_deg), int(k),
                              C.byref(out), MEM_HOST, C.byref(info)))
    np_ = int(info.n_pairs)
    return dict(local_idx=li[:np_].copy(), global_idx=gi[:np_].copy(), global_xyz=np.stack([gx[:np_], gy[:np_], gz[:np_]], 1),
                d2=d2[:np_].copy(), potential_pairings=int(info.potential_pairings))


def _pl_arrays(n):
    li = np.zeros(n, np.uint32)
    a = [np.zeros(n, np.float32) for _ in range(6)]
    return li, a, PairsPlOut(li.ctypes.data_as(_UP), *[x.ctypes.data_as(_FP) for x in a])


def nn_search_pt2pl(m: Map, s: Scan, T, distance_threshold, mode=PT2PL_PLANE_DISTANCE):
    """Matcher_Point2Plane on an NDT map (mh_nn_search_pt2pl)."""
    li, a, out = _pl_arrays(max(s.n, 1))
    info = MatchInfo()
    T = _T12(T)
    _chk(lib().mh_nn_search_pt2pl(m._h, s._h, T.ctypes.data_as(_DP), float(distance_threshold), int(mode), C.byref(out),
                                  MEM_HOST, C.byref(info)))
    k = int(info.n_pairs)
    return dict(local_idx=li[:k].copy(), centroid=np.stack(a[:3], 1)[:k].copy(), normal=np.stack(a[3:], 1)[:k].copy(),
                potential_pairings=int(info.potential_pairings))


def nn_search_pt2pl_knn(m: Map, s: Scan, T, distance_threshold, plane_eigen_threshold, search_radius, knn, minimum_plane_points):
    """Matcher_Point2Plane on a plain point map: k nearest neighbours + PCA (mh_nn_search_pt2pl_knn; rgbd.yaml:143-151)."""
    li, a, out = _pl_arrays(max(s.n, 1))
    info = MatchInfo()
    T = _T12(T)
    pr = Pt2PlKnnParams(float(distance_threshold), float(plane_eigen_threshold), float(search_radius), int(knn), int(minimum_plane_points))
    _chk(lib().mh_nn_search_pt2pl_knn(m._h, s._h, T.ctypes.data_as(_DP), C.byref(pr), C.byref(out), MEM_HOST, C.byref(info)))
    k = int(info.n_pairs)
    return dict(local_idx=li[:k].copy(), centroid=np.stack(a[:3], 1)[:k].copy(), normal=np.stack(a[3:], 1)[:k].copy(),
                potential_pairings=int(info.potential_pairings))


def icp_get_pt2pl_pairs(s: Scan):
    li, a, out = _pl_arrays(max(s.n, 1))
    k = C.c_uint64(0)
    _chk(lib().mh_icp_get_pt2pl_pairs(s._h, C.byref(out), MEM_HOST, C.byref(k)))
    k = int(k.value)
    return dict(local_idx=li[:k].copy(), centroid=np.stack(a[:3], 1)[:k].copy(), normal=np.stack(a[3:], 1)[:k].copy())


def nn_search_dense(m: Map, s: Scan, T):
    n = max(s.n, 1)
    gi = np.zeros(n, np.uint32)
    gx, gy, gz, d2 = (np.zeros(n, np.float32) for _ in range(4))
    T = _T12(T)
    _chk(lib().mh_nn_search_dense(m._h, s._h, T.ctypes.data_as(_DP), _vp(gi), _vp(gx), _vp(gy), _vp(gz), _vp(d2),
                                  MEM_HOST))
    k = s.n
    return dict(global_idx=gi[:k], global_xyz=np.stack([gx[:k], gy[:k], gz[:k]], 1), d2=d2[:k])


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
        return GNParamsC(self.max_inner_iterations, self.robust_kernel, self.robust_kernel_param, self.min_delta,
                         self.max_cost, self.weight_pt2pt, self.weight_pt2pl)


def _mk_prior(prior):
    if prior is None:
        return None
    mean, info = prior
    p = Prior()
    p.mean[:] = list(_T12(mean))
    p.info[:] = list(np.asarray(info, dtype=np.float64).reshape(36))
    return p


def _mk_pt2pt(local_xyz, global_xyz):
    arrs = list(_soa(local_xyz)) + list(_soa(global_xyz))
    return PairsPt2Pt(*[a.ctypes.data_as(_FP) for a in arrs], len(arrs[0])), arrs


def _mk_pt2pl(local_xyz, centroid_xyz, normal_xyz):
    arrs = list(_soa(local_xyz)) + list(_soa(centroid_xyz)) + list(_soa(normal_xyz))
    return PairsPt2Pl(*[a.ctypes.data_as(_FP) for a in arrs], len(arrs[0])), arrs


def gn_solve(ctx: Context, T, pt2pt=None, pt2pl=None, params: GNParams | None = None, prior=None):
    params = params or GNParams()
    gp = params.c()
    pp, k1 = _mk_pt2pt(*pt2pt) if pt2pt is not None else (None, None)
    pl, k2 = _mk_pt2pl(*pt2pl) if pt2pl is not None else (None, None)
    pr = _mk_prior(prior)
    Tio = _T12(T).copy()
    n_steps, ok = C.c_int32(0), C.c_int32(1)
    trace = (GNStep * max(1, params.max_inner_iterations))()
    _chk(lib().mh_gn_solve(ctx._h, C.byref(pp) if pp else None, C.byref(pl) if pl else None, MEM_HOST, C.byref(gp),
                           C.byref(pr) if pr else None, Tio.ctypes.data_as(_DP), C.byref(n_steps), C.byref(ok), trace))
    steps = [dict(H=np.array(s.H).reshape(6, 6), g=np.array(s.g), err_norm_sqr=s.err_norm_sqr, delta=np.array(s.delta),
                  T_after=np.array(s.T_after)) for s in trace]
    return Tio, int(n_steps.value), bool(ok.value), steps


def covariance(ctx: Context, T, pt2pt=None, pt2pl=None, findif_xyz=1e-7, findif_ang=1e-7):
    pp, k1 = _mk_pt2pt(*pt2pt) if pt2pt is not None else (None, None)
    pl, k2 = _mk_pt2pl(*pt2pl) if pt2pl is not None else (None, None)
    T = _T12(T)
    cov = np.zeros(36)
    _chk(lib().mh_covariance(ctx._h, C.byref(pp) if pp else None, C.byref(pl) if pl else None, MEM_HOST,
                             T.ctypes.data_as(_DP), findif_xyz, findif_ang, cov.ctypes.data_as(_DP)))
    return cov.reshape(6, 6)


@dataclass
class ICPParams:
    """mp2p_icp::Parameters + the matcher/solver settings of lidar3d-default.yaml:172-209."""
    max_iterations: int = 300
    min_abs_step_trans: float = 1e-4
    min_abs_step_rot: float = 5e-5
    disable_stall_test: bool = False
    threshold: object = None
    kernel_param: object = None
    threshold_angular_deg: float = 0.0
    pt2pl_threshold: object = None  # None, or per-iteration Matcher_Point2Plane.distanceThreshold (needs an NDT map)
    pt2pl_mode: int = 0             # PT2PL_PLANE_DISTANCE | PT2PL_CENTROID_DISTANCE
    matched_points: int = 0         # 0 = the point matcher pairs plane-paired points again, 1 = it skips them (U12)
    gn: GNParams = field(default_factory=GNParams)
    hook_enabled: bool = False
    hook_min_trans: float = 0.15
    hook_min_rot: float = float(np.deg2rad(0.75))
    hook_checkpoint: object = None
    compute_covariance: bool = True
    cov_findif_xyz: float = 1e-7
    cov_findif_ang: float = 1e-7
    poll_every: int = 0
    expected_iterations: int = 0
    profile: bool = False

    def c(self, T_guess):
        thr = np.ascontiguousarray(np.broadcast_to(np.asarray(self.threshold, np.float64), (self.max_iterations,)))
        kp = np.ascontiguousarray(np.broadcast_to(np.asarray(self.kernel_param, np.float64), (self.max_iterations,)))
        cp = ICPParamsC()
        cp.max_iterations = self.max_iterations
        cp.min_abs_step_trans = self.min_abs_step_trans
        cp.min_abs_step_rot = self.min_abs_step_rot
        cp.disable_stall_test = int(self.disable_stall_test)
        cp.threshold = thr.ctypes.data_as(_DP)
        cp.kernel_param = kp.ctypes.data_as(_DP)
        cp.threshold_angular_deg = self.threshold_angular_deg
        plt = None
        if self.pt2pl_threshold is not None:
            plt = np.ascontiguousarray(np.broadcast_to(np.asarray(self.pt2pl_threshold, np.float64), (self.max_iterations,)))
            cp.pt2pl_threshold = plt.ctypes.data_as(_DP)
        cp.gn = self.gn.c()
        cp.hook_enabled = int(self.hook_enabled)
        cp.hook_min_trans = self.hook_min_trans
        cp.hook_min_rot = self.hook_min_rot
        chk = _T12(self.hook_checkpoint if self.hook_checkpoint is not None else T_guess)
        cp.hook_checkpoint[:] = list(chk)
        cp.compute_covariance = int(self.compute_covariance)
        cp.cov_findif_xyz = self.cov_findif_xyz
        cp.cov_findif_ang = self.cov_findif_ang
        cp.poll_every = self.poll_every
        cp.pt2pl_mode = int(self.pt2pl_mode)
        cp.matched_points = int(self.matched_points)
        cp.expected_iterations = self.expected_iterations
        cp.profile = int(self.profile)  # 0 | 1 (all jobs) | 2 (job 0 of a batch only)
        return cp, (thr, kp, plt)


def _result_dict(res: ICPResult):
    return dict(T=np.array(res.T), cov=np.array(res.cov).reshape(6, 6), quality=res.quality,
                n_iterations=int(res.n_iterations), termination_reason=int(res.termination_reason),
                n_final_pairs=int(res.n_final_pairs), potential_pairings=int(res.potential_pairings),
                n_match_launches=int(res.n_match_launches), match_kernel_ms=res.match_kernel_ms, total_ms=res.total_ms,
                n_final_pairs_pt2pl=int(res.n_final_pairs_pt2pl), n_host_polls=int(res.n_host_polls),
                n_enqueued_iterations=int(res.n_enqueued_iterations))


def icp_align_prefers_solo(s: Scan, p: ICPParams, T_guess=None, concurrent_callers: int = 1) -> bool:
    """mh_icp_align_prefers_solo: would a single alignment of this scan run its loop in one launch?"""
    cp, keep = p.c(T_guess if T_guess is not None else np.eye(4))
    yes = C.c_int32(0)
    _chk(lib().mh_icp_align_prefers_solo(s._h, C.byref(cp), int(concurrent_callers), C.byref(yes)))
    return bool(yes.value)


def icp_align(m: Map, s: Scan, T_guess, p: ICPParams, prior=None, want_trace=True, want_pairs=False):
    cp, keep = p.c(T_guess)
    T0 = _T12(T_guess)
    res = ICPResult()
    trace = (ICPIter * max(1, p.max_iterations))() if want_trace else None
    pr = _mk_prior(prior)
    po = None
    if want_pairs:
        n = max(s.n, 1)
        li, gi = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        gx, gy, gz, d2 = (np.zeros(n, np.float32) for _ in range(4))
        po = PairsOut(li.ctypes.data_as(_UP), gi.ctypes.data_as(_UP), gx.ctypes.data_as(_FP), gy.ctypes.data_as(_FP),
                      gz.ctypes.data_as(_FP), d2.ctypes.data_as(_FP))
    _chk(lib().mh_icp_align(m._h, s._h, C.byref(cp), T0.ctypes.data_as(_DP), C.byref(pr) if pr else None, C.byref(res),
                            trace, C.byref(po) if po else None, MEM_HOST))
    out = _result_dict(res)
    if want_trace:
        n_it = out["n_iterations"]
        n_tr = min(p.max_iterations, n_it + 1)
        if TERM_NAMES[out["termination_reason"]] in ("NoPairings", "SolverError"):
            n_tr = n_it
        out["trace"] = [dict(T=np.array(trace[i].T), n_pairs=int(trace[i].n_pairs), threshold=trace[i].threshold,
                             kernel_param=trace[i].kernel_param, delta_trans=trace[i].delta_trans,
                             delta_rot=trace[i].delta_rot) for i in range(n_tr)]
    if want_pairs:
        k = out["n_final_pairs"] - out["n_final_pairs_pt2pl"]
        out["pairs"] = dict(local_idx=li[:k].copy(), global_idx=gi[:k].copy(),
                            global_xyz=np.stack([gx[:k], gy[:k], gz[:k]], 1), d2=d2[:k].copy())
    return out


def pairs_block_bytes(n_scan_points: int) -> int:
    return int(lib().mh_pairs_block_bytes(int(n_scan_points)))


def unpack_pairs_block(block: np.ndarray, scan_sizes, results):
    """Views of every job's final pairings inside a pairs block (uint8 array as filled by icp_align_batch)."""
    out, off = [], 0
    for n, r in zip(scan_sizes, results):
        nbytes = pairs_block_bytes(n)
        S = nbytes // 24
        k = r["n_final_pairs"] - r["n_final_pairs_pt2pl"]
        u = block[off:off + nbytes].view(np.uint32).reshape(6, S)
        f = block[off:off + nbytes].view(np.float32).reshape(6, S)
        out.append(dict(local_idx=u[0, :k], global_idx=u[1, :k], global_xyz=np.stack([f[2, :k], f[3, :k], f[4, :k]], 1),
                        d2=f[5, :k]))
        off += nbytes
    return out


class BatchCall:
    """The arguments of one mh_icp_align_batch call, marshalled once: a caller that repeats a batch (a replay, bench.py)
    keeps its argument arrays instead of rebuilding them per call -- per call that is the C function and nothing else.
    run() returns the raw result structs (ICPResult array, overwritten by the next run()); results() the usual dicts."""

    def __init__(self, maps, scans, T_guesses, p, priors=None, pairs_block=None, pairs_mem=MEM_HOST):
        n = self.n = len(scans)
        self._T = np.ascontiguousarray(np.stack([_T12(t) for t in T_guesses]).reshape(n * 12))
        if isinstance(p, (list, tuple)):  # one ICPParams per job (its own schedules, budget, hook check point)
            assert len(p) == n
            made = [q.c(self._T[12 * i:12 * i + 12]) for i, q in enumerate(p)]
            self._cp = (ICPParamsC * n)(*[m[0] for m in made])
            self._keep = [m[1] for m in made]
            self._cp_ref, self._per_job = self._cp, 1
        else:
            self._cp, self._keep = p.c(self._T[:12])
            self._cp_ref, self._per_job = C.byref(self._cp), 0
        self._maps, self._scans = list(maps), list(scans)  # (keep the handles' owners alive)
        self._mh = (C.c_void_p * n)(*[m._h for m in maps])
        self._sh = (C.c_void_p * n)(*[s._h for s in scans])
        self.res = (ICPResult * n)()
        self._pr_arr = None
        self._keep_pr = []
        if priors is not None:
            self._pr_arr = (C.POINTER(Prior) * n)()
            for i, pr in enumerate(priors):
                if pr is not None:
                    self._keep_pr.append(_mk_prior(pr))
                    self._pr_arr[i] = C.pointer(self._keep_pr[-1])
        self._block = pairs_block  # (a numpy array stays referenced)
        self._pb = None
        if pairs_block is not None:
            self._pb = C.c_void_p(pairs_block if isinstance(pairs_block, int) else pairs_block.ctypes.data)
        self._mem = pairs_mem
        self._fn = lib().mh_icp_align_batch
        self._Tp = self._T.ctypes.data_as(_DP)

    def run(self):
        _chk(self._fn(self.n, self._mh, self._sh, self._cp_ref, self._per_job, self._Tp, self._pr_arr, self.res, self._pb, self._mem))
        return self.res

    def results(self):
        return [_result_dict(r) for r in self.res]


def icp_align_batch(maps, scans, T_guesses, p: ICPParams, priors=None, pairs_block=None, pairs_mem=MEM_HOST):
    """One alignment per (map, scan) pair; every scan must live in its own Context (its own stream).
    pairs_block: None, a writable uint8 numpy array (host; pairs_mem MEM_HOST, or MEM_HOST_PINNED when its memory is
    page-locked -- the download then completes asynchronously, see molahip.h) or a raw pointer (int) with pairs_mem."""
    call = BatchCall(maps, scans, T_guesses, p, priors, pairs_block, pairs_mem)
    call.run()
    return call.results()


def preprocess_batch(raws, params, out_maps, out_icps=None):
    """mh_scan_preprocess_batch: the filter chain of several scans in one set of launches.  `params`: one PreprocessParams
    for all, or a list with one per scan; `out_icps`: None, or a list whose entries may be None."""
    n = len(raws)
    per_job = isinstance(params, (list, tuple))
    arr = (PreprocessParams * (n if per_job else 1))(*(params if per_job else [params]))
    hs = lambda scans: (C.c_void_p * n)(*[(sc._h if sc is not None else None) for sc in scans])
    _chk(lib().mh_scan_preprocess_batch(n, hs(raws), arr, C.sizeof(PreprocessParams) if per_job else 0, hs(out_maps),
                                        hs(out_icps) if out_icps is not None else None))


def preprocess_params(decim_map_resolution, decim_icp_resolution, min_points_to_filter=2000, index_mode=INDEX_FLOOR,
                      range_min=0.0, range_max=0.0, range_center=(0.0, 0.0, 0.0), bbox_mode=BBOX_OFF,
                      bbox_min=(0.0, 0.0, 0.0), bbox_max=(0.0, 0.0, 0.0), timestamp_method=TS_NONE,
                      time_offset=0.0, decim_map_method=DECIMATE_FIRST_POINT,
                      decim_icp_method=DECIMATE_FIRST_POINT) -> PreprocessParams:
    return PreprocessParams(float(decim_map_resolution), float(decim_icp_resolution), int(min_points_to_filter),
                            int(index_mode), float(range_min), float(range_max),
                            (C.c_float * 3)(*map(float, range_center)), int(bbox_mode),
                            (C.c_float * 3)(*map(float, bbox_min)), (C.c_float * 3)(*map(float, bbox_max)),
                            int(timestamp_method), float(time_offset), int(decim_map_method), int(decim_icp_method))
