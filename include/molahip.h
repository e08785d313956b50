/* molahip.h -- C ABI of libmolahip: the MI355X-native (gfx950 / HIP) implementation of the per-scan
 * ICP registration hot path that mola::LidarOdometry runs through mp2p_icp::ICP::align().
 *
 * This is the drop-in boundary.  No C++ or torch types cross it: opaque handles,
 * plain pointers + sizes, POD structs, integer status codes.  Every entry point names the reference
 * interface it stands in for ("file:line" is relative to /root/reference; [U] marks upstream classes
 * that the reference selects by name from its YAML but does not vendor -- SURVEY.md 0.1).
 * The C++ host layer (mola_lidar_odometry_amd/host/, namespace mp2p_icp_hip) and the mp2p_icp
 * plugin adapter are thin wrappers over exactly these functions.
 *
 * Conventions
 *  - Poses are `double T[12]`: row-major 3x4 [R|t] of "local wrt global" (the 3rd argument of
 *    ICP::align, LidarOdometry.cpp:961-962; mrpt::poses::CPose3D).
 *  - Tangent vectors / 6x6 matrices use [v(3); w(3)] ordering (LidarOdometry.cpp:977-984); the
 *    covariance is in (x,y,z,yaw,pitch,roll) like mrpt::poses::CPose3DPDFGaussian.
 *  - Point clouds are SoA float arrays (mrpt CPointsMap::getPointsBufferRef_{x,y,z} [U]).
 *  - `mem` arguments say where the caller's arrays live: MH_MEM_HOST (borrowed for the call, copied),
 *    MH_MEM_DEVICE (HIP device pointers on the context's device, read in place) or, where an entry point says
 *    so, MH_MEM_HOST_PINNED: page-locked host memory (hipHostMalloc / torch pin_memory) that the caller keeps
 *    valid and unmodified until the context's stream has passed the copy (mh_ctx_synchronize, or the return of
 *    the next blocking call that uses the scan) -- the copy is then asynchronous on the context's stream and
 *    the call returns at once, so uploads of the next scans overlap the alignment of the current ones.
 *  - Every function returns mh_status (0 = OK), never throws, never aborts; a message for the last
 *    failure on the calling thread is available from mh_last_error_string().
 *  - One context = one HIP device + one stream.  Contexts are independent and may be driven from
 *    different host threads; a single context (and the maps/scans created from it) must not be used
 *    from two threads at once -- the reference itself keeps one align() in flight per LidarOdometry
 *    instance (LidarOdometry.h:548-549, LidarOdometry.cpp:634).
 *  - There is NO CPU fallback: without a HIP device mh_ctx_create fails with MH_ERR_NO_DEVICE.
 */
#ifndef MOLAHIP_H
#define MOLAHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MH_API __attribute__((visibility("default")))

#define MH_VERSION_MAJOR 0
#define MH_VERSION_MINOR 1
#define MH_VERSION_PATCH 0
/* The parameter structs of this header carry no size field: they grow at the END, and every growth bumps MH_ABI_VERSION (6:
 * mh_preprocess_params' two decimation-method fields, round 5).  A binder built against this header checks
 * `mh_abi_version() == MH_ABI_VERSION` once after loading the library (capi.py does; the C++ host layer links the header it was
 * built with) and zero-initialises every struct it passes -- a field the binder does not know then reads as its default. */
#define MH_ABI_VERSION 6

typedef int32_t mh_status;
enum {
  MH_OK = 0,
  MH_ERR_INVALID_ARGUMENT = 1,
  MH_ERR_HIP = 2,            /* a HIP runtime call failed; see mh_last_error_string() */
  MH_ERR_OUT_OF_MEMORY = 3,
  MH_ERR_OUT_OF_RANGE = 4,   /* a voxel index does not fit the 21-bit-per-axis key */
  MH_ERR_NO_DEVICE = 5,
  MH_ERR_UNSUPPORTED = 6,
  MH_ERR_INTERNAL = 7,
  /* not a failure: the call has done what it was asked to; an EARLIER asynchronous call left something out (mh_map_insert) */
  MH_WARN_PREVIOUS_OUT_OF_RANGE = 64
};

enum { MH_MEM_HOST = 0, MH_MEM_DEVICE = 1, MH_MEM_HOST_PINNED = 2 };

/* coordinate -> voxel index rule (FLOOR is the default) */
enum { MH_INDEX_FLOOR = 0, MH_INDEX_TRUNC = 1 };

/* Voxel-index distance used by remove_voxels_farther_than (lidar3d-default.yaml:237-238; the comment there says "L1",
 * upstream's code is unverified [U]): max(|dk|) (default), |dkx|+|dky|+|dkz|, or sqrt(dkx^2+dky^2+dkz^2), each
 * compared with ceil(remove_voxels_farther_than / voxel_size).  A run-time switch until the reference decides it. */
enum { MH_FAR_CHEBYSHEV = 0, MH_FAR_L1 = 1, MH_FAR_L2 = 2 };

/* mp2p_icp::RobustKernel [U] as selected at lidar3d-default.yaml:188.  The exact upstream form of
 * GemanMcClure is unverified (SURVEY App.B U1), hence the variants. */
enum {
  MH_KERNEL_NONE = 0,
  MH_KERNEL_GM_C4 = 1,     /* w = c^4/(c^2+e^2)^2  (default) */
  MH_KERNEL_GM_KISS = 2,   /* w = c^2/(c+e^2)^2 */
  MH_KERNEL_GM_BARRON = 3, /* w = 1/(e^2/(4c^2)+1)^2 */
  MH_KERNEL_CAUCHY = 4,    /* w = c^2/(c^2+e^2) */
  MH_KERNEL_GM_C2 = 5      /* w = c^2/(c^2+e^2)^2 */
};

/* mp2p_icp::IterTermReason [U] (used in-tree at LidarOdometry.cpp:970,1007,1019) */
enum {
  MH_TERM_UNDEFINED = 0,
  MH_TERM_NO_PAIRINGS = 1,
  MH_TERM_SOLVER_ERROR = 2,
  MH_TERM_MAX_ITERATIONS = 3,
  MH_TERM_STALLED = 4,
  MH_TERM_QUALITY_CHECKPOINT_FAILED = 5,
  MH_TERM_HOOK_REQUEST = 6
};

typedef struct mh_ctx mh_ctx;
typedef struct mh_map mh_map;
typedef struct mh_scan mh_scan;

/* ------------------------------------------------------------------------------------------------
 * Library / context
 * ---------------------------------------------------------------------------------------------- */
MH_API mh_status mh_version(uint32_t* major, uint32_t* minor, uint32_t* patch);
MH_API const char* mh_last_error_string(void);
/* MH_ABI_VERSION of the loaded library (see above). */
MH_API uint32_t mh_abi_version(void);
/* A status at or above MH_WARN_PREVIOUS_OUT_OF_RANGE is not a failure: the call did its work (MH_SUCCEEDED is the test a binder
 * wants where it would write `== MH_OK`). */
#define MH_SUCCEEDED(status) ((status) == MH_OK || (status) >= MH_WARN_PREVIOUS_OUT_OF_RANGE)
MH_API const char* mh_status_string(mh_status s);
MH_API mh_status mh_device_count(int32_t* n);

/* `hip_stream`: a hipStream_t to run on (e.g. the caller's torch stream), or NULL to let the context
 * create and own a non-blocking stream. */
MH_API mh_status mh_ctx_create(int32_t device, void* hip_stream, mh_ctx** out);
MH_API mh_status mh_ctx_destroy(mh_ctx* ctx);
MH_API mh_status mh_ctx_synchronize(mh_ctx* ctx);
MH_API mh_status mh_ctx_stream(mh_ctx* ctx, void** hip_stream_out);
/* Free / total device memory [bytes] of the context's GPU after draining its stream (capacity planning for batches of
 * maps and scans; also how the tests check that destroyed handles give their memory back). */
MH_API mh_status mh_ctx_memory_info(mh_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes);

/* Page-locked host memory for MH_MEM_HOST_PINNED uploads (hipHostMalloc / hipHostFree behind the C ABI, for host code
 * that does not link the HIP runtime itself). */
MH_API mh_status mh_host_alloc_pinned(size_t bytes, void** out);
MH_API mh_status mh_host_free_pinned(void* p);

/* ------------------------------------------------------------------------------------------------
 * Local map: the NN-search target.  Replaces mola::HashedVoxelPointCloud [U] as configured at
 * lidar3d-default.yaml:228-242 (creationOpts.voxel_size :233, insertOpts.max_points_per_voxel :235)
 * in its role as mrpt::maps::NearestNeighborsCapable [U] for the matcher.
 * Device layout: open-addressing hash table of 16-byte slots {packed voxel key, first, count} plus
 * voxel-contiguous 16-byte point records {x,y,z,source index}; see DESIGN.md.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  float voxel_size;              /* [m] > 0 */
  uint32_t max_points_per_voxel; /* 0 = unlimited */
  uint32_t index_mode;           /* MH_INDEX_* */
  /* mola::NDT [U] role (lidar3d-ndt.yaml:236-254); all 0 = plain HashedVoxelPointCloud */
  float min_distance_between_points; /* insertOpts: drop a point closer than this to a stored point of its voxel */
  float ndt_max_eigen_ratio;         /* insertOpts.max_eigen_ratio_for_planes; > 0 enables per-voxel NDT statistics */
  uint32_t ndt_min_points;           /* voxels with fewer stored points carry no NDT (0 -> 4) */
  uint32_t far_voxel_metric;         /* MH_FAR_* : how mh_map_insert measures "farther than" (see there) */
} mh_map_params;

typedef struct {
  uint64_t n_points;   /* stored points (after the per-voxel cap) */
  uint64_t n_offered;  /* points offered to the last build */
  uint64_t n_voxels;   /* occupied voxels */
  uint64_t table_size; /* hash slots (power of two) */
  float bbox_min[3], bbox_max[3];
  float voxel_size;
  uint32_t max_points_per_voxel;
  uint64_t n_planes;   /* voxels whose NDT is a plane (0 when NDT statistics are off) */
  uint32_t deferred_status; /* MH_OK, or the verdict of the last mh_map_insert that no call has reported yet (see there) */
  uint32_t reserved_;
} mh_map_info;

MH_API mh_status mh_map_create(mh_ctx* ctx, const mh_map_params* params, mh_map** out);
MH_API mh_status mh_map_destroy(mh_map* map);
/* (Re)build from n points: equivalent to clear() + insertPoint() for each point in order
 * (HashedVoxelPointCloud::insertPoint [U] via FilterMerge, lidar3d-default.yaml:362-368): a point whose
 * voxel already holds max_points_per_voxel is dropped; non-finite points are dropped.  The "global
 * index" reported by the NN search is the point's index in these arrays. */
/* Testing hook (fault injection; no effect on results): the next `first_attempts` device allocations of the library's grow-only
 * buffers report out-of-memory on their first attempt (the library then returns its retired blocks to the runtime and asks for
 * exactly what it needs), the next `retries` of those second attempts fail as well (the call then returns MH_ERR_OUT_OF_MEMORY
 * and leaves every handle usable).  Process-wide counters. */
MH_API mh_status mh_debug_fail_allocations(int32_t first_attempts, int32_t retries);
MH_API mh_status mh_map_build(mh_map* map, const float* x, const float* y, const float* z, size_t n, int32_t mem);
MH_API mh_status mh_map_get_info(const mh_map* map, mh_map_info* info);
/* Incremental key-frame update, device resident.  Replaces FilterMerge ->
 * HashedVoxelPointCloud::insertPointCloud [U] (lidar3d-default.yaml:362-368, LidarOdometry.cpp:1161-1206) for the
 * layer `scan` (vehicle frame, input_layer_in_local_coordinates: true): every point is composed with the robot
 * pose T (row-major 3x4, fp64, result rounded to float) and offered to insertPoint in order, after everything the
 * map already stores; then, if remove_voxels_farther_than > 0 (insertOpts, yaml:238), every voxel whose index
 * distance (mh_map_params::far_voxel_metric; default max(|dkx|,|dky|,|dkz|)) to the voxel of T's translation exceeds
 * ceil(remove_voxels_farther_than/voxel_size) is erased [U].  The source index of a new point is (points ever offered to this map) + its index in `scan`.
 * Nothing travels to the host except four counters, and those lazily: the call returns once the update is QUEUED.
 * Deferred verdict: points whose voxel index leaves the +-2^20 range of the packed key are left out, and that is known
 * only when the counters arrive.  It is reported -- once, as MH_WARN_PREVIOUS_OUT_OF_RANGE, a status of its own that is NOT a
 * failure -- by the NEXT mh_map_insert on this map, which performs its own insertion as always (a caller can tell "inserted" from
 * "not inserted": every MH_ERR_* means not inserted, this one and MH_OK mean inserted; the wrappers log it and go on); until then
 * mh_map_get_info shows it in mh_map_info::deferred_status (as MH_ERR_OUT_OF_RANGE).  mh_map_get_info and the downloads
 * never fail for it.  (mh_map_build is synchronous about it: it builds the map without the offending points, sets
 * n_offered, and returns MH_ERR_OUT_OF_RANGE itself.) */
MH_API mh_status mh_map_insert(mh_map* map, const mh_scan* scan, const double T[12], float remove_voxels_farther_than);
/* Copy the stored content to HOST arrays (any may be NULL): points voxel by voxel, voxels in ascending
 * (kx,ky,kz), in-voxel insertion order.  xyz/src_idx hold n_points entries, vox_* hold n_voxels. */
MH_API mh_status mh_map_download(const mh_map* map, float* x, float* y, float* z, uint32_t* src_idx,
                                 int32_t* vox_keys_xyz, uint32_t* vox_first, uint32_t* vox_count);
/* NDT statistics per occupied voxel, same voxel order as mh_map_download (HOST arrays of n_voxels entries, any may be
 * NULL): centroid, unit normal (largest component positive) and the plane flag.  Zeros when NDT is off. */
MH_API mh_status mh_map_download_ndt(const mh_map* map, float* cx, float* cy, float* cz, float* nx, float* ny, float* nz,
                                     uint32_t* is_plane);

/* ------------------------------------------------------------------------------------------------
 * Scan: the local point layer handed to align() ("decimated_for_icp", lidar3d-default.yaml:204),
 * untransformed, in the vehicle frame.
 * ---------------------------------------------------------------------------------------------- */
MH_API mh_status mh_scan_create(mh_ctx* ctx, const float* x, const float* y, const float* z, size_t n, int32_t mem,
                                mh_scan** out);
/* Replace the points (e.g. after the caller re-ran its de-skew, LidarOdometry.cpp:992-999).  `mem` may be
 * MH_MEM_HOST_PINNED (asynchronous upload, see Conventions). */
MH_API mh_status mh_scan_update(mh_scan* scan, const float* x, const float* y, const float* z, size_t n, int32_t mem);
/* Replace the points from an interleaved buffer, the form raw sensor data arrives in: point i has float32 x/y/z at
 * data + i*point_step + off_{x,y,z} and, with off_t >= 0, a float32 time stamp [s] at off_t (a KITTI velodyne .bin is
 * point_step 16 / offsets 0,4,8; a sensor_msgs/PointCloud2 payload gives its own).  This is the step the reference's
 * observations_generator (mp2p_icp_filters::Generator, lidar3d-default.yaml:250-262) performs on the CPU when it turns
 * the raw observation into the SoA 'raw' layer; here: ONE copy of the bytes and a de-interleave kernel.  point_step and
 * the offsets are multiples of 4.  With off_t < 0 the scan carries no time stamps afterwards.  `mem` may be
 * MH_MEM_HOST_PINNED (asynchronous upload, see Conventions). */
MH_API mh_status mh_scan_update_aos(mh_scan* scan, const void* data, size_t n, size_t point_step, size_t off_x,
                                    size_t off_y, size_t off_z, int64_t off_t, int32_t mem);
/* Optional: queue the construction of the scan's search order for the tile matcher (large layers: the points sorted by
 * 2x2x2-voxel block of the local frame and cut into tiles) right behind an upload, for the voxel size of the
 * map it will be aligned against, so that it overlaps whatever else the device is doing.  Asynchronous on the context's
 * stream.  mh_icp_align / mh_icp_align_batch build it themselves when it is missing or stale. */
MH_API mh_status mh_scan_prepare(const mh_scan* scan, float voxel_size);
MH_API mh_status mh_scan_destroy(mh_scan* scan);
MH_API mh_status mh_scan_size(const mh_scan* scan, uint64_t* n);

/* ------------------------------------------------------------------------------------------------
 * Scan pre-processing on the device: the observation filter chain that produces the
 * layers `decimated_for_map` / `decimated_for_icp` from the raw sensor cloud.  Replaces, for the
 * configuration of lidar3d-default.yaml:270-350, mp2p_icp_filters::{FilterAdjustTimestamps,
 * FilterDecimateVoxels(FirstPoint | ClosestToAverage), FilterByRange, FilterBoundingBox, FilterDeskew} [U].
 * ---------------------------------------------------------------------------------------------- */
enum { MH_TS_NONE = 0, MH_TS_MIDDLE_IS_ZERO = 1, MH_TS_EARLIEST_IS_ZERO = 2 }; /* TimestampAdjustMethod (yaml:275) */
enum { MH_BBOX_OFF = 0, MH_BBOX_KEEP_OUTSIDE = 1, MH_BBOX_KEEP_INSIDE = 2 };
/* FilterDecimateVoxels decimate_method [U]: FirstPoint (the shipped lidar3d pipelines, yaml:291) keeps the first point of every
 * voxel in input order; ClosestToAverage (rgbd.yaml:254-278; the commented alternative at lidar3d-default.yaml:292) keeps, per
 * voxel, the point closest to the voxel's mean -- mean = (float sum of the voxel's points in input order) * (1.0f / count),
 * squared error (dx*dx + dy*dy) + dz*dz in float, the first of equally close points. */
enum { MH_DECIMATE_FIRST_POINT = 0, MH_DECIMATE_CLOSEST_TO_AVERAGE = 1 };

typedef struct {
  float decim_map_resolution;    /* FilterDecimateVoxels #1 voxel_filter_resolution (yaml:289); 0 = stage skipped */
  float decim_icp_resolution;    /* FilterDecimateVoxels #2 (yaml:316); 0 = stage skipped */
  uint32_t min_points_to_filter; /* minimum_input_points_to_filter (yaml:290,317): smaller inputs pass undecimated */
  int32_t index_mode;            /* MH_INDEX_FLOOR | MH_INDEX_TRUNC of the decimation grid */
  float range_min, range_max;    /* FilterByRange (yaml:301-302), inclusive; range_max <= 0 = stage skipped */
  float range_center[3];
  int32_t bbox_mode;             /* FilterBoundingBox (yaml:305-310): MH_BBOX_* ; the pipeline keeps the OUTSIDE */
  float bbox_min[3], bbox_max[3];
  int32_t timestamp_method;      /* FilterAdjustTimestamps (yaml:270-276): MH_TS_* ; ignored without time stamps */
  float time_offset;
  int32_t decim_map_method;      /* decimate_method of FilterDecimateVoxels #1: MH_DECIMATE_* (0 = FirstPoint) */
  int32_t decim_icp_method;      /* ... of FilterDecimateVoxels #2 */
} mh_preprocess_params;

/* Attach per-point time stamps [s] (relative to the scan's reference time) to a scan of the same size. */
MH_API mh_status mh_scan_set_timestamps(mh_scan* scan, const float* t, size_t n, int32_t mem);
/* raw -> decimate(map res) -> by-range -> bounding box -> out_map -> decimate(icp res) -> out_icp (may be NULL).
 * Survivors keep the raw order (upstream's FirstPoint decimation emits them in the iteration order of its hash
 * container, which is implementation-defined: same set, deterministic order here).  The outputs carry adjusted
 * time stamps (when `raw` has them) and each point's index in `raw`; they are the *_skewed layers. */
MH_API mh_status mh_scan_preprocess(const mh_scan* raw, const mh_preprocess_params* params, mh_scan* out_map,
                                    mh_scan* out_icp);
/* The same chain for the scans of several sequences at once: every step is ONE launch over all of them, issued on the
 * stream of the first scan's context (work still queued on the other scans' streams is waited for), and one read-back
 * brings all counts -- N sequences in one process cost the host one sequence's launches (the callers' threads contend
 * for the runtime otherwise).  `params` is an array with `params_stride` bytes between entries (0: one set for all);
 * `out_icps` may be NULL, or hold NULL entries.  Outputs are those of n_jobs separate mh_scan_preprocess calls, bit for
 * bit.  Each job's three scans belong to one context; the contexts may differ from job to job (same device). */
MH_API mh_status mh_scan_preprocess_batch(size_t n_jobs, const mh_scan* const* raws, const mh_preprocess_params* params,
                                          size_t params_stride, mh_scan* const* out_maps, mh_scan* const* out_icps);
/* FilterDeskew (yaml:328-350): p' = Exp_SO3(w*t_i)*p + v*t_i with twist = (vx,vy,vz,wx,wy,wz) in the vehicle frame,
 * fp64, rounded to float [U].  twist == NULL or a scan without time stamps copies the points (skip_deskew /
 * silently_ignore_no_timestamps).  `out` must differ from `in`; it is what align() and mh_map_insert() consume, and
 * `in` stays valid for the re-de-skew inside the ICP loop (LidarOdometry.cpp:992-999) with no host round trip.
 * `in` may belong to another context of the same device (a layer prepared on a second stream whose work the caller
 * has synchronised); the kernel is ordered on `out`'s stream. */
MH_API mh_status mh_scan_deskew(const mh_scan* in, const double twist[6], mh_scan* out);
/* Both layers of a scan (`a`: the large one for the map, `b`: the small one for the ICP) de-skewed by ONE launch, which
 * also leaves the bounding box of the de-skewed `b` -- the input of the sensor-range low-pass that runs next
 * (LidarOdometry.cpp:744, 1515-1545) -- in host memory: one launch and one wait on the per-scan chain instead of
 * mh_scan_deskew x 2 + mh_scan_bbox.  Same results bit for bit; falls back to those calls for copies (twist NULL or no
 * time stamps), empty layers and a `b` above 65536 points. */
MH_API mh_status mh_scan_deskew_pair(const mh_scan* in_a, const mh_scan* in_b, const double twist[6], mh_scan* out_a,
                                     mh_scan* out_b, float bb_min[3], float bb_max[3], uint64_t* n_finite);
/* Axis-aligned bounding box of the finite points (CPointsMap::boundingBox [U], used by the sensor-range estimate at
 * LidarOdometry.cpp:1503-1508, 1517-1534).  n_finite (nullable) = number of finite points; zeros for an empty scan. */
MH_API mh_status mh_scan_bbox(const mh_scan* scan, float bb_min[3], float bb_max[3], uint64_t* n_finite);
/* Copy a scan to HOST arrays (any may be NULL; t / src_idx are zero-filled when the scan has none). */
MH_API mh_status mh_scan_download(const mh_scan* scan, float* x, float* y, float* z, float* t, uint32_t* src_idx);

/* ------------------------------------------------------------------------------------------------
 * Matcher-granular path.  Replaces mp2p_icp::Matcher_Points_DistanceThreshold::implMatchOneLayer [U]
 * (lidar3d-default.yaml:195-204; pairingsPerPoint 1, allowMatchAlreadyMatchedGlobalPoints true) on
 * top of NearestNeighborsCapable::nn_single_search [U]: p' = (float)(R*l+t); NN over the 3x3x3 voxel
 * block around voxel(p'); accepted iff d^2 < thr^2 + ang^2*|p'|^2.  Output = mp2p_icp::Pairings::
 * paired_pt2pt [U] as SoA, compacted in ascending local index.  Output arrays hold scan-size entries,
 * live in `mem`, and any of them may be NULL.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  uint32_t* local_idx;
  uint32_t* global_idx;
  float* gx;
  float* gy;
  float* gz;
  float* d2; /* errorSquareAfterTransformation */
} mh_pairs_out;

typedef struct {
  uint64_t n_pairs;
  uint64_t potential_pairings; /* Pairings::potential_pairings [U] = scan size * pairingsPerPoint */
} mh_match_info;

MH_API mh_status mh_nn_search(const mh_map* map, const mh_scan* scan, const double T[12], double threshold,
                              double threshold_angular_deg, const mh_pairs_out* out, int32_t mem,
                              mh_match_info* info);
/* The same matcher with pairingsPerPoint = k > 1 (rgbd.yaml:135-141) on NearestNeighborsCapable::nn_multiple_search [U]
 * ("same scan keeping k best sorted"): per local point the k nearest map points of the 27-voxel block in
 * ascending (d^2, scan position), accepted in that order while d^2 < thr^2 + ang^2*|p'|^2.  Pairs in ascending local index, a
 * point's pairs in ascending distance; output arrays hold scan-size * k entries; potential_pairings = scan size * k.
 * pairings_per_point 1 is mh_nn_search. */
#define MH_MAX_PAIRINGS_PER_POINT 8
MH_API mh_status mh_nn_search_k(const mh_map* map, const mh_scan* scan, const double T[12], double threshold,
                                double threshold_angular_deg, uint32_t pairings_per_point, const mh_pairs_out* out,
                                int32_t mem, mh_match_info* info);
/* Un-compacted variant: one entry per scan point; global_idx = 0xFFFFFFFF where nothing was found in
 * the 27-voxel block (no threshold applied).  Arrays hold scan-size entries; any may be NULL. */
MH_API mh_status mh_nn_search_dense(const mh_map* map, const mh_scan* scan, const double T[12], uint32_t* global_idx,
                                    float* gx, float* gy, float* gz, float* d2, int32_t mem);

/* Replaces mp2p_icp::Matcher_Point2Plane::implMatchOneLayer [U] on a mola::NDT [U] map (lidar3d-ndt.yaml:195-200,
 * 236-254).  The upstream semantics are unverified (SURVEY App.B U10); implemented default: among
 * the planar voxels of the 3x3x3 block around voxel(p') the one with the nearest centroid is taken (fp32 d^2, first
 * minimum in scan order) and the pairing {centroid, normal, local point} is emitted iff |n.(p'-c)| < distance_threshold.
 * Needs a map built with ndt_max_eigen_ratio > 0.  Output = Pairings::paired_pt2pl [U] as SoA, ascending local index. */
typedef struct {
  uint32_t* local_idx;
  float *cx, *cy, *cz; /* plane centroid */
  float *nx, *ny, *nz; /* plane unit normal */
} mh_pairs_pl_out;

/* What Matcher_Point2Plane.distanceThreshold (lidar3d-ndt.yaml:197) is compared with -- SURVEY App.B U10, a run-time
 * switch until the reference decides it: the point-to-plane distance |n.(p'-c)| (default) or the distance to the
 * plane's centroid |p'-c| (fp32 squares, un-fused).  The search for the nearest planar voxel is the same. */
enum { MH_PT2PL_PLANE_DISTANCE = 0, MH_PT2PL_CENTROID_DISTANCE = 1 };
/* Matcher_Points_Base::allowMatchAlreadyMatchedPoints [U] (default false upstream, not set by either target pipeline): a
 * matcher skips the local points an earlier matcher of the same iteration has paired.  In lidar3d-ndt.yaml:195-210 that keeps
 * plane-paired points out of Matcher_Points_DistanceThreshold.  Unverified (U12), hence a switch; PAIR_AGAIN is what rounds
 * 1-3 did and stays the default until the reference decides (MOLA_HIP_MATCHED_POINTS=skip|again in the host layers). */
enum { MH_MATCHED_POINTS_PAIR_AGAIN = 0, MH_MATCHED_POINTS_SKIP = 1 };

MH_API mh_status mh_nn_search_pt2pl(const mh_map* map, const mh_scan* scan, const double T[12], double distance_threshold,
                                    uint32_t mode, const mh_pairs_pl_out* out, int32_t mem, mh_match_info* info);
/* The same matcher on a map WITHOUT plane statistics -- a mola::HashedVoxelPointCloud layer, /root/reference/pipelines/rgbd.yaml:143-151
 * (distanceThreshold 0.40, planeEigenThreshold 1e-2, searchRadius 0.80, knn 10, minimumPlanePoints 6); SURVEY 8a row a13 "otherwise
 * KNN + PCA" [U]: per transformed local point the knn nearest map points of the 27-voxel block (nn_multiple_search [U]), those
 * with d^2 < searchRadius^2 (a prefix: ascending distances), none if fewer than max(3, minimumPlanePoints); mean + covariance of
 * them (fp64), eigenvalues e0 <= e1 <= e2; a plane iff e2 > 0 and e0 <= planeEigenThreshold * e2; normal = unit eigenvector of e0
 * (sign: largest component positive -- the residual and its Jacobian are even in n); pairing {centroid, normal, local point} iff
 * |n.(p'-c)| <= distanceThreshold.  Output as mh_nn_search_pt2pl; works on any map (the NDT statistics, if any, are not used). */
#define MH_MAX_PLANE_KNN 16
typedef struct {
  double distance_threshold;      /* [m] */
  double plane_eigen_threshold;   /* e0 / e2 */
  double search_radius;           /* [m] */
  uint32_t knn;                   /* 3 .. MH_MAX_PLANE_KNN */
  uint32_t minimum_plane_points;  /* >= 3 */
} mh_pt2pl_knn_params;
MH_API mh_status mh_nn_search_pt2pl_knn(const mh_map* map, const mh_scan* scan, const double T[12], const mh_pt2pl_knn_params* params,
                                        const mh_pairs_pl_out* out, int32_t mem, mh_match_info* info);

/* ------------------------------------------------------------------------------------------------
 * Solver-granular path.  Replaces mp2p_icp::Solver_GaussNewton::impl_optimal_pose /
 * optimal_tf_gauss_newton [U] (lidar3d-default.yaml:184-190): robust-weighted point-to-point (3-row)
 * and point-to-plane (1-row) terms, optional prior factor (LidarOdometry.cpp:859-875), 6x6 LDL^T,
 * T <- T (+) exp(delta), repeated max_inner_iterations times on the same pairings.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const float *lx, *ly, *lz; /* local (untransformed) */
  const float *gx, *gy, *gz; /* global */
  size_t n;
} mh_pairs_pt2pt;

typedef struct {
  const float *lx, *ly, *lz; /* local point */
  const float *cx, *cy, *cz; /* plane centroid */
  const float *nx, *ny, *nz; /* plane unit normal */
  size_t n;
} mh_pairs_pt2pl;

typedef struct {
  double mean[12]; /* prior pose (CPose3DPDFGaussianInf::mean) */
  double info[36]; /* 6x6 information matrix, row-major (cov_inv) */
} mh_prior;

typedef struct {
  uint32_t max_inner_iterations; /* Solver_GaussNewton maxIterations (yaml:187) */
  uint32_t robust_kernel;        /* MH_KERNEL_* (yaml:188) */
  double robust_kernel_param;    /* yaml:190 */
  double min_delta;              /* inner early exit, 1e-7 */
  double max_cost;               /* "target error" early exit, 0 */
  double weight_pt2pt;           /* 1.0 */
  double weight_pt2pl;           /* 1.0 */
} mh_gn_params;

typedef struct {
  double H[36];
  double g[6];
  double err_norm_sqr;
  double delta[6];
  double T_after[12];
} mh_gn_step;

/* T_io: linearisation point in (SolverContext::guessRelativePose [U]), solution out.
 * `n_steps` receives the number of solves performed; `trace` (nullable) holds max_inner_iterations
 * entries.  Returns MH_OK with *solver_ok = 0 when the 6x6 solve produced non-finite values. */
MH_API mh_status mh_gn_solve(mh_ctx* ctx, const mh_pairs_pt2pt* pt2pt, const mh_pairs_pt2pl* pt2pl, int32_t mem,
                             const mh_gn_params* params, const mh_prior* prior, double T_io[12], int32_t* n_steps,
                             int32_t* solver_ok, mh_gn_step* trace);

/* Replaces mp2p_icp::covariance() [U] (result consumed at LidarOdometry.cpp:1009,1035-1036,2090):
 * central-difference Jacobian of the stacked residuals wrt (x,y,z,yaw,pitch,roll), cov = (A^T A)^-1;
 * diag(1e6) when there are no pairings or A^T A is singular. */
MH_API mh_status mh_covariance(mh_ctx* ctx, const mh_pairs_pt2pt* pt2pt, const mh_pairs_pt2pl* pt2pl, int32_t mem,
                               const double T[12], double findif_xyz, double findif_ang, double cov[36]);

/* ------------------------------------------------------------------------------------------------
 * Fused path.  Replaces mp2p_icp::ICP::align [U] as called at LidarOdometry.cpp:961-962 with the
 * pipeline of lidar3d-default.yaml:162-209 (one Matcher_Points_DistanceThreshold, one
 * Solver_GaussNewton, QualityEvaluator_PairedRatio).  The whole loop -- match, accumulate, 6x6 solve,
 * stall test, hook test, quality, covariance -- runs on the device; the host only enqueues kernels and
 * polls a termination flag every `poll_every` iterations.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  uint32_t max_iterations;      /* mp2p_icp::Parameters::maxIterations (yaml:173) */
  double min_abs_step_trans;    /* yaml:174 */
  double min_abs_step_rot;      /* yaml:175 */
  uint32_t disable_stall_test;  /* 1: run exactly max_iterations (BASELINE configs[1]) */
  /* Per-iteration values of the run-time formulas of yaml:198 (matcher threshold) and yaml:190
   * (robust kernel parameter), indexed by ICP_ITERATION; HOST arrays of max_iterations entries. */
  const double* threshold;
  const double* kernel_param;
  double threshold_angular_deg; /* yaml:200 */
  /* NULL, or max_iterations values of Matcher_Point2Plane.distanceThreshold (lidar3d-ndt.yaml:197): that matcher
   * then runs before the point matcher in every iteration and both pairing sets go to one solve (ndt yaml:195-210). */
  const double* pt2pl_threshold;
  mh_gn_params gn;              /* robust_kernel_param ignored (kernel_param[k] is used) */
  /* Device-side equivalent of the in-tree iteration hook (LidarOdometry.cpp:923-952): request a stop
   * when the pose has moved more than hook_min_trans [m] or hook_min_rot [rad] from hook_checkpoint. */
  uint32_t hook_enabled;
  double hook_min_trans;
  double hook_min_rot;
  double hook_checkpoint[12];
  uint32_t compute_covariance;
  double cov_findif_xyz;        /* 1e-7 */
  double cov_findif_ang;        /* 1e-7 */
  uint32_t poll_every;          /* ICP iterations enqueued between host polls of the done flag; 0 = automatic: the
                                   first chunk as long as the context's previous alignment of the same kind ran (a fresh
                                   call and a re-entry after a hook request are predicted separately, see
                                   expected_iterations), then short ones */
  uint32_t expected_iterations; /* with poll_every = 0: the caller's own estimate of how many iterations this call will
                                   run (e.g. what its previous call of the same kind ran), 0 = let the library predict */
  uint32_t pt2pl_mode;          /* MH_PT2PL_* : the acceptance test of the point-to-plane matcher (with pt2pl_threshold) */
  uint32_t matched_points;      /* MH_MATCHED_POINTS_* : what the point matcher does with local points that the point-to-plane
                                   matcher of the same iteration has paired (with pt2pt_threshold) */
  uint32_t profile;             /* 1: time every match kernel with HIP events on the context stream (such a job is
                                   enqueued kernel by kernel instead of replaying the captured graph); 2: in
                                   mh_icp_align_batch, do that for job 0 only (lock step: its share of the launches) */
} mh_icp_params;

typedef struct {
  double T[12];
  uint32_t n_pairs;
  double threshold;
  double kernel_param;
  double delta_trans; /* |log(T_prev^-1 T_new)| translation part */
  double delta_rot;
} mh_icp_iter;

typedef struct {
  double T[12];                /* Results::optimal_tf.mean */
  double cov[36];              /* Results::optimal_tf.cov (x,y,z,yaw,pitch,roll) */
  double quality;              /* Results::quality (PairedRatio) */
  uint32_t n_iterations;       /* Results::nIterations */
  uint32_t termination_reason; /* Results::terminationReason, MH_TERM_* */
  uint32_t n_final_pairs;
  uint64_t potential_pairings;
  /* profile == 1 only: */
  uint32_t n_match_launches;
  double match_kernel_ms;      /* sum of the match kernel durations */
  double total_ms;             /* stream time of the whole align */
  uint32_t n_final_pairs_pt2pl; /* how many of n_final_pairs are point-to-plane */
  /* host-side bookkeeping of the device loop (what a latency budget wants to know): */
  uint32_t n_host_polls;          /* times the host waited for the device loop (1 = the first chunk was long enough) */
  uint32_t n_enqueued_iterations; /* iterations worth of kernels enqueued; those beyond the executed ones were early-exit launches */
} mh_icp_result;

/* `trace` (nullable, HOST, max_iterations entries) receives one record per executed iteration;
 * `final_pairs` (nullable, arrays in `pairs_mem`, scan-size entries) receives Results::finalPairings. */
MH_API mh_status mh_icp_align(const mh_map* map, const mh_scan* scan, const mh_icp_params* params,
                              const double T_guess[12], const mh_prior* prior, mh_icp_result* result,
                              mh_icp_iter* trace, const mh_pairs_out* final_pairs, int32_t pairs_mem);

/* Scheduling hint for callers that merge the alignments of several sequences into mh_icp_align_batch calls: *yes = 1 when a
 * single mh_icp_align of this scan with these parameters would run its whole loop in ONE kernel launch (layers of at most 2560
 * points under automatic loop control, see mh_debug_loop_stats) AND the loops of `concurrent_callers` such callers fit into
 * 70 % of the device's CUs together (four 1400-point layers on 256 CUs; the callers' other stages run beside the loops).  Such an alignment is better issued on its own at once than held
 * back for a lock-step batch (4 sequences of the default pipeline: 4450 against 3620 scans/s); with more callers than fit,
 * lock-step batches are faster (8 sequences: 4950 against 4100).  Results do not depend on the choice. */
MH_API mh_status mh_icp_align_prefers_solo(const mh_scan* scan, const mh_icp_params* params, uint32_t concurrent_callers,
                                           int32_t* yes);

/* Statistics (process-wide, no effect on results): single alignments of small layers (<= 2560 points) run their whole loop
 * in ONE kernel launch whose workgroups exchange partial sums among themselves, as long as the workgroups of all such loops
 * running on the device fit its CUs; `loops_started` counts them, `loops_abandoned` those whose workgroups gave up waiting for
 * each other and that were run again launch by launch (same result bit for bit; expected to stay 0).  Either may be NULL. */
MH_API void mh_debug_loop_stats(uint64_t* loops_started, uint64_t* loops_abandoned);

/* 1 when the library was built with -DMH_DEV_VARIANTS (tools/build_variants.sh): the matcher families that were measured against
 * the product kernels and lost -- MH_MATCH=t (tile matcher, map records staged in LDS), w (wave matcher), o (sorted scan) -- are
 * then selectable for A/B runs and parity tests.  The shipped library returns 0 and rejects those three values of MH_MATCH. */
MH_API int32_t mh_debug_dev_variants(void);

/* Results::finalPairings.paired_pt2pl [U] of the LAST mh_icp_align run on `scan`'s context (arrays in `mem`, scan-size
 * entries, any may be NULL). */
MH_API mh_status mh_icp_get_pt2pl_pairs(const mh_scan* scan, const mh_pairs_pl_out* out, int32_t mem, uint64_t* n_pairs);

/* Many independent alignments from one host thread, one context per job.  Job i uses maps[i], scans[i] (distinct
 * contexts), params[i] when params_per_job != 0 (else the one *params: N sequences have N adaptive thresholds, iteration
 * budgets and hook check points), guesses + 12*i, priors[i] (array or entries may be NULL), and writes results[i]; every
 * result is bitwise what mh_icp_align gives for that job alone.  Jobs that run the same kernel chain advance in LOCK
 * STEP: each kernel of an iteration is one launch over all of them (the jobs' tails fill each other's idle lanes) --
 * large layers (quad / tile matcher), 8-12 k-point layers (row matcher with the fused accumulation), layers up to 8 k
 * points (k_step16: search + sums per launch, the Gauss-Newton step carried into the next launch; also with
 * Matcher_Point2Plane on NDT maps); the rest is interleaved, one stream per job.  With profile = 2, job 0's match_kernel_ms is its share of the lock-step match
 * launches.  Work still queued on the jobs' own streams (asynchronous uploads, de-skew, filters) is ordered before the
 * batch, whichever stream the batch runs on.
 *
 * `pairs_block` (nullable): Results::finalPairings of every job.  Job i's part starts at byte offset
 * sum_{j<i} mh_pairs_block_bytes(scan size of job j) and holds six arrays of S = mh_pairs_block_bytes(n)/24 entries
 * each -- local_idx | global_idx (uint32) | gx | gy | gz | d2 (float) -- of which the first
 * results[i].n_final_pairs - results[i].n_final_pairs_pt2pl are valid, ascending local index.  `pairs_mem`:
 * MH_MEM_DEVICE, MH_MEM_HOST, or MH_MEM_HOST_PINNED: the download is then queued on a copy stream of the first job's
 * context and the call returns without waiting for it (it overlaps the next batch); mh_ctx_synchronize(scans[0]'s
 * context) waits for it, and so does the next batch before it overwrites the device-side staging. */
MH_API size_t mh_pairs_block_bytes(size_t n_scan_points);
MH_API mh_status mh_icp_align_batch(size_t n_jobs, const mh_map* const* maps, const mh_scan* const* scans,
                                    const mh_icp_params* params, int32_t params_per_job, const double* T_guesses,
                                    const mh_prior* const* priors, mh_icp_result* results, void* pairs_block,
                                    int32_t pairs_mem);

#ifdef __cplusplus
}
#endif
#endif /* MOLAHIP_H */
