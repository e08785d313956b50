/* icp_oracle.h -- CPU restatement ("oracle") of the per-scan ICP registration hot path
 * that mola::LidarOdometry drives through mp2p_icp::ICP::align().
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it -- as the checker / the timed CPU baseline, never as
 * the thing shipped.  The product path (libmolahip.so) neither links nor calls anything here.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in third-party packages that are not
 * vendored in /root/reference and are not pinned anywhere (mp2p_icp, mola_metric_maps, mrpt;
 * SURVEY.md section 0 and 8c), the reference cannot be built in this image, and its only tests
 * for the path are two end-to-end replays whose inputs are external.  This file therefore
 * restates the published algorithm of those packages (mp2p_icp ~1.6.x, mola_metric_maps ~1.2.x,
 * MRPT 2.13/2.14, Oct 2024) as specified in SURVEY.md section 8(a) + Appendix A, anchored on the
 * in-tree call sites cited per function below.  It is cross-checked by an independent float64
 * numpy restatement (oracle/icp_oracle_np.py) and by analytical known-answer tests.
 * Every upstream behaviour that could not be verified is a run-time switch.
 *
 * All citations "file:line" are relative to /root/reference.
 */
#ifndef ICP_ORACLE_H
#define ICP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums (values shared with include/molahip.h by convention, not by #include) ---- */
enum { ORC_INDEX_FLOOR = 0, ORC_INDEX_TRUNC = 1 };
enum {
  ORC_KERNEL_NONE = 0,
  ORC_KERNEL_GM_C4 = 1,    /* w = c^4/(c^2+e^2)^2     (SURVEY App.B U1 (i), default) */
  ORC_KERNEL_GM_KISS = 2,  /* w = c^2/(c+e^2)^2       (U1 (ii))                     */
  ORC_KERNEL_GM_BARRON = 3,/* w = 1/(e^2/(4c^2)+1)^2  (U1 (iii))                    */
  ORC_KERNEL_CAUCHY = 4,   /* w = c^2/(c^2+e^2)                                      */
  ORC_KERNEL_GM_C2 = 5     /* w = c^2/(c^2+e^2)^2     (un-normalised GM)             */
};
/* mp2p_icp::IterTermReason (in-tree users: LidarOdometry.cpp:970,1007,1019) */
enum {
  ORC_TERM_UNDEFINED = 0,
  ORC_TERM_NO_PAIRINGS = 1,
  ORC_TERM_SOLVER_ERROR = 2,
  ORC_TERM_MAX_ITERATIONS = 3,
  ORC_TERM_STALLED = 4,
  ORC_TERM_QUALITY_CHECKPOINT_FAILED = 5,
  ORC_TERM_HOOK_REQUEST = 6
};

/* ---- SE(3) helpers (mrpt::poses::CPose3D / Lie::SE<3>) -------------
 * A pose is T[12] = row-major 3x4 [R | t].  Tangent ordering [v(3); w(3)], translation first
 * (LidarOdometry.cpp:977-984).  ypr is MRPT's TPose3D (x,y,z,yaw,pitch,roll), R = Rz*Ry*Rx
 * (LidarOdometry.cpp:235; lidar3d-default.yaml:368). */
void orc_pose_from_ypr(const double xyzypr[6], double T[12]);
void orc_pose_to_ypr(const double T[12], double xyzypr[6]);
void orc_pose_compose(const double A[12], const double B[12], double AB[12]); /* A (+) B */
void orc_pose_inverse(const double A[12], double Ainv[12]);
void orc_se3_exp(const double xi[6], double T[12]);
void orc_se3_log(const double T[12], double xi[6]);
void orc_so3_log(const double T[12], double w[3]);

/* ---- local map: mola::HashedVoxelPointCloud (lidar3d-default.yaml:228-242) ------------ */
typedef struct orc_map orc_map;
typedef struct {
  float voxel_size;              /* creationOpts.voxel_size            (yaml:233) */
  uint32_t max_points_per_voxel; /* insertOpts.max_points_per_voxel    (yaml:235); 0 = no cap */
  uint32_t index_mode;           /* ORC_INDEX_FLOOR (SURVEY App.A) | ORC_INDEX_TRUNC */
  /* mola::NDT role (lidar3d-ndt.yaml:236-254): */
  float min_distance_between_points; /* insertOpts (ndt yaml:244): drop a point closer than this to a stored point of its voxel; 0 = off */
  float ndt_max_eigen_ratio;         /* max_eigen_ratio_for_planes (ndt yaml:248); 0 = no NDT statistics */
  uint32_t ndt_min_points;           /* voxels with fewer points have no NDT (4) */
  uint32_t far_voxel_metric;         /* remove_voxels_farther_than's voxel-index distance (yaml:237-238; the comment there says
                                        "L1", upstream's code is unverified): 0 = max|dk| (default), 1 = sum|dk|, 2 = Euclid */
} orc_map_params;

orc_map* orc_map_create(const orc_map_params* p);
void orc_map_destroy(orc_map* m);
/* insertPoint for n points in order; a point whose voxel already holds the cap is dropped.
 * Non-finite points are dropped.  The "global index" of a stored point is its position in the
 * concatenation of everything ever offered to orc_map_insert (its source index). */
void orc_map_insert(orc_map* m, const float* x, const float* y, const float* z, size_t n);
size_t orc_map_num_points(const orc_map* m);
size_t orc_map_num_voxels(const orc_map* m);
void orc_map_bbox(const orc_map* m, float mn[3], float mx[3]);
/* Dump the stored points voxel by voxel, voxels ordered by (kx,ky,kz) ascending, in-voxel
 * insertion order.  Arrays must hold orc_map_num_points() / orc_map_num_voxels() entries.
 * Any pointer may be NULL. */
void orc_map_dump(const orc_map* m, float* x, float* y, float* z, uint32_t* src_idx,
                  int32_t* vox_keys /*[3*V]*/, uint32_t* vox_first /*[V]*/, uint32_t* vox_count /*[V]*/);

/* NearestNeighborsCapable::nn_single_search on the 3x3x3 voxel block around voxel(q)
 * (SURVEY 8a row a8): x outer / y middle / z inner, in-voxel insertion order, strict '<' keeps
 * the first minimum.  Returns 1 if something was found.  n_candidates (may be NULL) is
 * incremented by the number of distance evaluations, n_voxels_hit by non-empty voxels visited. */
int orc_map_nn_single(const orc_map* m, float qx, float qy, float qz, float out_pt[3], float* out_d2,
                      uint32_t* out_src_idx, uint64_t* n_candidates, uint64_t* n_voxels_hit);

/* ---- matcher: mp2p_icp::Matcher_Points_DistanceThreshold (yaml:195-204), pairingsPerPoint=1,
 * allowMatchAlreadyMatchedGlobalPoints=true.  Local points are transformed p' = (float)(R*l+t)
 * with double pose; accepted iff d^2 < thr^2 + ang^2 * |p'|^2 (strict).  Pairs are emitted in
 * ascending local index.  Returns the number of pairs.  Output arrays sized n. */
typedef struct {
  uint64_t potential_pairings;
  uint64_t n_candidates;  /* distance evaluations (for the P-bar of SURVEY 8d) */
  uint64_t n_voxels_hit;
} orc_match_stats;
size_t orc_match_points(const orc_map* m, const float* lx, const float* ly, const float* lz, size_t n,
                        const double T[12], double threshold, double threshold_angular_deg,
                        uint32_t* local_idx, uint32_t* global_idx, float* gx, float* gy, float* gz,
                        float* d2, orc_match_stats* stats, int n_threads);

/* ---- the same matcher with pairingsPerPoint = k > 1 (rgbd.yaml:135-141) on NearestNeighborsCapable::nn_multiple_search [U]
 * ("same scan keeping k best sorted"): the k nearest of the 3x3x3 block in ascending (d^2, scan position),
 * accepted in that order while d^2 < thr^2 + ang^2 * |p'|^2.  Pairs in ascending local index, a point's pairs in ascending
 * distance.  Output arrays sized n * k; potential_pairings = n * k. */
int orc_map_nn_multiple(const orc_map* m, float qx, float qy, float qz, uint32_t k, float* out_pts, float* out_d2,
                        uint32_t* out_src_idx);
size_t orc_match_points_k(const orc_map* m, const float* lx, const float* ly, const float* lz, size_t n, const double T[12],
                          double threshold, double threshold_angular_deg, uint32_t k, uint32_t* local_idx,
                          uint32_t* global_idx, float* gx, float* gy, float* gz, float* d2, orc_match_stats* stats);

/* ---- matcher: mp2p_icp::Matcher_Point2Plane on a mola::NDT map (lidar3d-ndt.yaml:195-200; SURVEY 8a row a13,
 * App.B U10 -- the upstream semantics are unverified, this is the documented default): per voxel with
 * >= ndt_min_points points: mean, covariance (1/(n-1)), eigen-decomposition; the voxel is a plane iff
 * lambda_min/lambda_max < ndt_max_eigen_ratio, normal = eigenvector of lambda_min (sign: largest component > 0).
 * For a transformed local point p' the planar voxel of the 3x3x3 block with the nearest centroid (fp32 d^2, first
 * minimum in scan order) is taken and the pairing {centroid, normal, local point} is emitted iff
 * |n.(p'-c)| < distanceThreshold.  Pairs in ascending local index; arrays sized n. */
size_t orc_match_pt2pl(const orc_map* m, const float* lx, const float* ly, const float* lz, size_t n, const double T[12],
                       double distance_threshold, uint32_t* local_idx, float* cx, float* cy, float* cz, float* nx,
                       float* ny, float* nz, int n_threads);
/* ---- the same matcher on a map WITHOUT plane statistics (pipelines/rgbd.yaml:143-151: a HashedVoxelPointCloud layer; SURVEY 8a
 * row a13 "otherwise KNN + PCA") [U] -- restated from the parameter list of the reference's own pipeline (distanceThreshold,
 * planeEigenThreshold, searchRadius, knn, minimumPlanePoints) and the documented reading: per transformed local point p'
 *   nn_multiple_search(p', knn) (the k nearest of the 3x3x3 block, ascending (d^2, scan position));
 *   the neighbours with d^2 < searchRadius^2 (fp32; ascending distances: a prefix); fewer than max(3, minimumPlanePoints) -> none;
 *   their mean and covariance (fp64, 1/(m-1)), eigenvalues e0 <= e1 <= e2 (cyclic Jacobi, as the NDT statistics);
 *   a plane iff e2 > 0 and e0 <= planeEigenThreshold * e2;  normal = eigenvector of e0 (unit; sign: largest component > 0 -- the
 *   residual n.(p-c) and its Jacobian are even in n);  pairing {centroid, normal, local point} iff |n.(p'-c)| <= distanceThreshold
 *   (fp64).  Pairs in ascending local index; arrays sized n. */
typedef struct {
  double distance_threshold, plane_eigen_threshold, search_radius;
  uint32_t knn, minimum_plane_points;
} orc_pt2pl_knn_params;
size_t orc_match_pt2pl_knn(const orc_map* m, const float* lx, const float* ly, const float* lz, size_t n, const double T[12],
                           const orc_pt2pl_knn_params* p, uint32_t* local_idx, float* cx, float* cy, float* cz, float* nx,
                           float* ny, float* nz);
/* NDT statistics of the occupied voxels in ascending key order (same order as orc_map_dump): centroid, normal,
 * is_plane flag; arrays of orc_map_num_voxels() entries. */
void orc_map_dump_ndt(const orc_map* m, float* cx, float* cy, float* cz, float* nx, float* ny, float* nz,
                      uint32_t* is_plane);

/* ---- solver: mp2p_icp::Solver_GaussNewton / optimal_tf_gauss_newton (yaml:184-190) ------ */
typedef struct {
  const float *lx, *ly, *lz; /* local point, untransformed */
  const float *gx, *gy, *gz; /* global point */
  size_t n;
} orc_pairs_pt2pt;
typedef struct {
  const float *lx, *ly, *lz;    /* local point */
  const float *cx, *cy, *cz;    /* plane centroid */
  const float *nx, *ny, *nz;    /* unit normal */
  size_t n;
} orc_pairs_pt2pl;
typedef struct {
  double mean[12]; /* T_prior */
  double info[36]; /* 6x6 information, [v;w] ordering (LidarOdometry.cpp:859-875) */
} orc_prior;
typedef struct {
  uint32_t max_inner_iterations; /* Solver_GaussNewton.maxIterations (yaml:187) */
  uint32_t robust_kernel;        /* ORC_KERNEL_* (yaml:188) */
  double robust_kernel_param;    /* c (yaml:190) */
  double min_delta;              /* 1e-7 (U8) */
  double max_cost;               /* 0 (U8) */
  double weight_pt2pt, weight_pt2pl; /* pair weights, 1.0 */
} orc_gn_params;
/* Optional trace of each inner step: H (6x6 row-major), g (6), cost, delta (6). */
typedef struct {
  double H[36];
  double g[6];
  double err_norm_sqr;
  double delta[6];
  double T_after[12];
} orc_gn_step;
/* Returns the number of inner steps executed (solves done).  T_io: linearisation point in,
 * solution out. */
int orc_gn_solve(const orc_pairs_pt2pt* pp, const orc_pairs_pt2pl* pl, const orc_gn_params* p,
                 const orc_prior* prior /*nullable*/, double T_io[12], orc_gn_step* trace /*nullable,
                 [max_inner_iterations]*/, int n_threads);

/* mp2p_icp::covariance: numeric Jacobian (central differences) of the stacked
 * residual vector w.r.t. (x,y,z,yaw,pitch,roll); cov = (A^T A)^-1.  No pairings: diag(1e6). */
void orc_covariance(const orc_pairs_pt2pt* pp, const orc_pairs_pt2pl* pl, const double T[12],
                    double findif_xyz, double findif_ang, double cov[36], double AtA[36] /*nullable*/);

/* ---- ICP::align (called at LidarOdometry.cpp:961-962) -------------- */
typedef struct {
  uint32_t max_iterations;     /* yaml:173 */
  double min_abs_step_trans;   /* yaml:174 */
  double min_abs_step_rot;     /* yaml:175 */
  uint32_t disable_stall_test; /* C2 of SURVEY 8d runs exactly max_iterations */
  /* matcher */
  const double* threshold;     /* [max_iterations] value of the yaml:198 formula per ICP_ITERATION */
  double threshold_angular_deg;/* yaml:200 */
  const double* pt2pl_threshold; /* NULL = no Matcher_Point2Plane; else [max_iterations] distanceThreshold (ndt yaml:197),
                                    that matcher runs BEFORE the point matcher and both pairing sets go to one solve */
  /* solver */
  const double* kernel_param;  /* [max_iterations] value of the yaml:190 formula */
  orc_gn_params gn;            /* robust_kernel_param ignored (taken from kernel_param[k]) */
  /* iteration hook emulation (LidarOdometry.cpp:923-952): stop when the pose moved more than
   * hook_min_trans / hook_min_rot (rad) away from hook_checkpoint. */
  uint32_t hook_enabled;
  double hook_min_trans, hook_min_rot;
  double hook_checkpoint[12];
  /* covariance */
  uint32_t compute_covariance;
  double cov_findif_xyz, cov_findif_ang;
  /* Matcher_Points_Base::allowMatchAlreadyMatchedPoints [U] (added in round 4 as U12): upstream's matchers skip
   * local points that an EARLIER matcher of the same iteration has paired unless that parameter is true (default false, and
   * neither target pipeline sets it).  In the NDT pipeline (lidar3d-ndt.yaml:195-210: Matcher_Point2Plane, then
   * Matcher_Points_DistanceThreshold on the same layer) that keeps plane-paired points out of the point-to-point matcher.
   * 0 = every matcher pairs every point (rounds 1-3), 1 = points with a plane pairing get no point pairing. */
  uint32_t pt2pt_skip_plane_paired;
} orc_icp_params;

typedef struct {
  double T[12];
  uint32_t n_pairs;
  double threshold, kernel_param;
  double delta_trans, delta_rot;
} orc_icp_iter;

typedef struct {
  double T[12];
  double cov[36];
  double quality;
  uint32_t n_iterations;
  uint32_t termination_reason;
  uint32_t n_final_pairs;      /* point-to-point + point-to-plane */
  uint64_t potential_pairings;
  uint64_t n_candidates_total; /* sum over iterations, for P-bar */
  uint32_t n_final_pairs_pt2pl;
} orc_icp_result;

/* final pairings (optional): arrays sized n_local */
typedef struct {
  uint32_t* local_idx;
  uint32_t* global_idx;
  float *gx, *gy, *gz, *d2;
} orc_pairs_out;

int orc_icp_align(const orc_map* m, const float* lx, const float* ly, const float* lz, size_t n,
                  const double T_guess[12], const orc_icp_params* p, const orc_prior* prior,
                  orc_icp_result* res, orc_icp_iter* trace /*nullable [max_iterations]*/,
                  orc_pairs_out* final_pairs /*nullable*/, int n_threads);

/* ======================================================================================
 * scan pre-processing and the incrementally updated local map.
 * The filter classes live in mp2p_icp_filters / mola_metric_maps [U] (not vendored): restated
 * from their published behaviour, anchored on the YAML that instantiates them.
 * ==================================================================================== */

/* HashedVoxelPointCloud::insertPointCloud via FilterMerge (lidar3d-default.yaml:362-368,
 * input_layer_in_local_coordinates: true): every point is composed with the robot pose
 * (CPose3D::composePoint, fp64, rounded to float) and offered to insertPoint in order; then,
 * when remove_voxels_farther_than > 0 (yaml:238), every voxel whose index distance
 * max(|dkx|,|dky|,|dkz|) to the voxel of the insertion pose exceeds
 * ceil(remove_voxels_farther_than / voxel_size) is erased [U]. */
void orc_map_insert_posed(orc_map* m, const float* x, const float* y, const float* z, size_t n, const double T[12],
                          float remove_voxels_farther_than);

/* FilterAdjustTimestamps (yaml:270-276): method 1 = MiddleIsZero (t -= (tmin+tmax)/2),
 * 2 = EarliestIsZero (t -= tmin); then + time_offset.  float arithmetic [U].  In place. */
enum { ORC_TS_NONE = 0, ORC_TS_MIDDLE_IS_ZERO = 1, ORC_TS_EARLIEST_IS_ZERO = 2 };
void orc_adjust_timestamps(float* t, size_t n, int method, float time_offset);

/* FilterDecimateVoxels, DecimateMethod::FirstPoint (yaml:285-292, 312-319): the first point (input order)
 * that falls into each voxel of size `resolution` survives; an input smaller than min_points_to_filter is passed
 * through.  out_idx receives the surviving input indices in ASCENDING order (upstream emits them in the iteration
 * order of its hash container, which is implementation-defined; the set is the same).  Non-finite points are
 * dropped.  Returns the number of survivors. */
size_t orc_decimate_first_point(const float* x, const float* y, const float* z, size_t n, float resolution,
                                uint32_t min_points_to_filter, int index_mode, uint32_t* out_idx);

/* FilterDecimateVoxels, DecimateMethod::ClosestToAverage [U] (pipelines/rgbd.yaml:254-278; the commented alternative of
 * lidar3d-default.yaml:292): per voxel, mean = (float sum of the voxel's points in input order) * (1.0f / count); the point
 * with the smallest (dx*dx + dy*dy) + dz*dz to it survives, a strictly smaller error replacing the candidate (the first of
 * equally close points stays).  Pass-through below min_points_to_filter, non-finite points dropped, ascending output as for
 * FirstPoint.  mp2p_icp_filters is not vendored: the float accumulation and the first-of-equals rule are this restatement's
 * reading; tests/test_oracle_decimate.py pins it against an independent numpy reading. */
enum { ORC_DECIMATE_FIRST_POINT = 0, ORC_DECIMATE_CLOSEST_TO_AVERAGE = 1 };
size_t orc_decimate_closest_to_average(const float* x, const float* y, const float* z, size_t n, float resolution,
                                       uint32_t min_points_to_filter, int index_mode, uint32_t* out_idx);

/* FilterByRange (yaml:297-302): keep range_min^2 <= |p-center|^2 <= range_max^2, float arithmetic.
 * FilterBoundingBox (yaml:305-310): inside = min <= p <= max on every axis; keep_inside selects which side is
 * emitted (the default pipeline keeps the OUTSIDE, `outside_pointcloud_layer`).  Index lists, ascending. */
size_t orc_filter_by_range(const float* x, const float* y, const float* z, size_t n, float range_min, float range_max,
                           const float center[3], uint32_t* out_idx);
size_t orc_filter_bbox(const float* x, const float* y, const float* z, size_t n, const float bb_min[3],
                       const float bb_max[3], int keep_inside, uint32_t* out_idx);

/* FilterDeskew (yaml:328-350): with the constant twist (vx,vy,vz,wx,wy,wz) of the vehicle frame, every point is
 * moved by the pose reached after its own time stamp: p' = Exp_SO3(w*t_i) * p + v*t_i  (fp64, rounded to float) [U]. */
void orc_deskew(const float* x, const float* y, const float* z, const float* t, size_t n, const double twist[6],
                float* ox, float* oy, float* oz);

/* The 1st-pass chain of lidar3d-default.yaml:278-319 on one raw scan:
 * decimate(res_map) -> by-range -> bounding box -> [map layer] -> decimate(res_icp) -> [icp layer].
 * idx_map / idx_icp (each sized n) receive indices into the RAW scan, ascending. */
typedef struct {
  float decim_map_resolution, decim_icp_resolution; /* 0 = stage skipped */
  uint32_t min_points_to_filter;
  int32_t index_mode;
  float range_min, range_max; /* range_max <= 0: FilterByRange skipped */
  float range_center[3];
  int32_t bbox_mode; /* 0 skipped, 1 keep outside, 2 keep inside */
  float bbox_min[3], bbox_max[3];
  int32_t decim_map_method, decim_icp_method; /* ORC_DECIMATE_* of the two decimations */
} orc_preprocess_params;
void orc_preprocess(const float* x, const float* y, const float* z, size_t n, const orc_preprocess_params* p,
                    uint32_t* idx_map, size_t* n_map, uint32_t* idx_icp, size_t* n_icp);

int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
