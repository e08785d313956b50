// mh_icp.hip -- the hot path: correspondence search, residual/Jacobian accumulation, 6x6 Gauss-Newton
// solve and the ICP outer loop, all on the device (gfx950).
//
// Stands in for (all [U] = upstream classes the reference selects by name):
//   mp2p_icp::ICP::align                                   <- LidarOdometry.cpp:961-962
//   mp2p_icp::Matcher_Points_DistanceThreshold              <- lidar3d-default.yaml:195-204
//   mola::HashedVoxelPointCloud::nn_single_search           <- lidar3d-default.yaml:228-242
//   mp2p_icp::Solver_GaussNewton / optimal_tf_gauss_newton  <- lidar3d-default.yaml:184-190
//   mp2p_icp::QualityEvaluator_PairedRatio                  <- lidar3d-default.yaml:206-209
//   mp2p_icp::covariance                                    <- LidarOdometry.cpp:1009
//
// Kernel sequence per ICP iteration (everything stays in HBM/L2; the host never sees pairings):
//   k_match4         a DPP quad per scan point: transform, exact branch-and-bound NN over the 27-voxel block,
//                    threshold test, store pairing            (MH_MATCH=p|x: k_match<fused>, one lane per point,
//                    with the first accumulation fused in)
//   k_accum          robust weight + 18 fp64 moment sums of the stored pairings at the current pose -> partials
//   k_solve          one workgroup: ordered sum of partials, prior factor, LDL^T, T <- T(+)exp(delta),
//                    inner/outer loop bookkeeping, stall + hook tests, termination flag, next threshold
//   k_accum, k_solve (inner Gauss-Newton steps >= 1 on the SAME pairings)
// Smaller layers take shorter chains (chosen by size in AlignJob::start / enqueue_chunk):
//   <= 32 k points   k_match16: a DPP row (16 lanes) per point; up to 12 k points it also accumulates the first step
//                    (k_match16<.,true> | k_solve | k_accum | k_solve)
//   <= 8 k points    k_step16 x max_inner: search + sums per launch, the Gauss-Newton step carried into the next launch -- what
//                    lidar3d-default.yaml's 1-3 k-point ICP layer runs, alone and in lock-step batches (profiled alignments
//                    keep a match kernel of their own: the chain above)
//   NDT maps         Matcher_Point2Plane rides in the row kernels (k_step16<true>, k_match16<true,.>), its Gauss-Newton rows are
//                    summed alongside (k_accum_both); above 32 k points k_match_pl (one lane per point)
// A chunk of iterations is one hipGraph launch when its shape repeats (direct launches otherwise); every kernel begins
// with `if (st->done) return`.
// No fp atomics anywhere: reductions are fixed-shape trees, so results are bitwise reproducible.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <new>
#include <type_traits>
#include <vector>

#include "mh_nn_device.h"
#include "mh_nn_flat.h"

#ifndef MH_LOOPW_DEFAULT
#define MH_LOOPW_DEFAULT "batch"
#endif

using namespace mh;

constexpr uint32_t kBlock = 256;
#ifndef MH_QUAD_WAVES
#define MH_QUAD_WAVES 8  // waves per SIMD the register allocator has to make room for in the quad kernel: 64 VGPRs, which
                         // MH_QUAD_W = 4 records in flight per lane fit without scratch (mh_nn_device.h has the sweep)
#endif
#ifndef MH_ACCUM_WAVES
#define MH_ACCUM_WAVES 1  // min waves per SIMD asked of the register allocator for k_accum (tuning knob)
#endif
#ifndef MH_MATCH_WAVES
#define MH_MATCH_WAVES 1  // min waves per SIMD asked of the register allocator for k_match (tuning knob)
#endif
constexpr uint32_t kMaxGnTrace = 16;

struct IcpDeviceState {
  double T[12];
  double T_prev[12];
  uint32_t iter, inner, done, term_reason;
  uint32_t n_pairs, n_iterations, solver_ok, n_solves;
  uint32_t cov_done, n_pairs_pl;
  float cur_thr2, cur_ang2;  // matcher threshold^2 of iteration `iter` and the angular term: k_match4 reads nothing but this block
  uint32_t pending, serial;  // k_step16: the partials of a Gauss-Newton step wait for their solve; (alignment's epoch << 22) + its launches so far
  double cur_kparam;         // robust-kernel parameter of iteration `iter` (no dependent table look-up in k_accum*)
  double cov[36];
  double covD[72];  // (T(x+h_j) - T(x-h_j)) / (2 h_j), j = 0..5, 3x4 each
  uint32_t handover_timeouts, pad2_;  // k_step16: workgroups that gave up waiting for the state / the partials they expected (never seen)
  uint32_t dbg[8];  // the first give-up: [0] 1 = state, 2 = partials' tag  [1] workgroup  [2] thread  [3] wanted  [4] seen  [5] groups  [6] seen B
};

// Per-alignment parameters live in DEVICE memory (uploaded once per align from a pinned host mirror) and the kernels
// receive pointers to them: the kernel arguments of a whole chunk of iterations are then identical from one alignment
// to the next, so the chunk can be captured once into a hipGraph and replayed with one host call instead of ~80 launches.
struct MatchK {
  const double* thr;     // [max_iterations] device
  const double* kparam;  // [max_iterations] device
  float ang2;
  uint32_t kernel;
  double w_pt2pt;
  double kparam_fixed;   // solver-granular path: fixed robust-kernel parameter
  uint32_t use_fixed;
  uint32_t skip_pl_paired;  // MH_MATCHED_POINTS_SKIP: a point with a point-to-plane pairing gets no point pairing (U12)
  const double* pl_thr;  // [max_iterations] device: Matcher_Point2Plane.distanceThreshold per iteration (or null)
  double w_pt2pl;
};

struct SolveK {
  uint32_t max_iterations, disable_stall, max_inner, has_prior;
  double min_step_trans, min_step_rot, min_delta, max_cost;
  uint32_t hook_enabled, pad;
  double hook_trans, hook_rot;
  double hook_chk_inv[12];
  double prior_mean_inv[12];
  double prior_info[36];
  const double* thr;
  const double* kparam;
  mh_icp_iter* trace;
  mh_gn_step* gn_trace;
  double cov_hx, cov_ha;  // finite-difference steps of the covariance
  // streaming loop control (AlignJob::run_streaming): a word of page-locked HOST memory (device-visible address) that the
  // kernel closing a Gauss-Newton step updates with (ICP iteration about to run | done << 31); null = not published
  uint32_t* host_progress;
  // k_step16 launches replayed from a captured graph (frozen arguments) are told their place IN the chunk; the serial number
  // the chunk starts from is written here by the host before every replay (no launch skips the check)
  uint32_t step_base, step_pad;
};

struct IcpDeviceParams {
  MatchK mk;
  SolveK sk;
};

struct PoseArg {
  double m[12];
};

// mh_icp_align_batch, lock-step mode: one descriptor per alignment; the *_b kernels take blockIdx.y as the job index.
// One launch over all jobs keeps the device full across their tails: per 120 k-point scan the match step costs 9.7 us
// in a launch of sixteen scans' worth of points, 21.8 us alone (tools/batch_hypothesis.py).
struct BatchJob {
  IcpDeviceState* st;
  IcpDeviceState* st_b;  // k_step16_b: the other half of the state ping-pong
  uint32_t serial_base, serial_pad;  // ... and the serial number its uploaded state block carries
  const MatchK* mk;
  const SolveK* sk;
  const float *lx, *ly, *lz;
  uint32_t n, nb, nba, nbm;  // nbm: columns of the partials the FIRST solve of an iteration reads (who wrote them)
  MapView map;
  float4* pair_q;
  uint32_t* pair_gidx;
  double* part;
  // final pairings of the batch (mh_icp_align_batch's pairs_block), or null
  double* partb;                 // point-to-plane partials (NDT chains) or null
  float4 *pl_c, *pl_n;           // point-to-plane pairings or null
  uint32_t* sched_dst;           // where this job keeps its threshold schedules (staged start-up copy) ...
  uint32_t sched_dwords, stage_off;  // ... their size, and where this job's [state | params | schedules] start in the staging block
  // tile matcher: the scan in search order (mh_tile.hip)
  const float *sx, *sy, *sz;
  const uint32_t *perm, *tile_start;
  uint32_t n_tiles, tile_pad;
  uint32_t* cp_counts;   // [nb] pairs per 256-point block | [nb] exclusive offsets
  uint32_t* cp_out;      // six arrays of cp_stride entries: local_idx | global_idx | gx | gy | gz | d2
  uint32_t cp_stride, cp_pad;
  // k_icp16_b: this job's exchange block (16-byte entries: point-to-point sums | point-to-plane sums) and the serial number its entries start from
  void *loop_xa, *loop_xb;
  uint32_t loop_serial0, loop_pad;
};

// LDS hand-off between lanes of ONE wave: LDS operations of a wave execute in order, so only the compiler has to be
// kept from moving the accesses across this point.
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// Workgroup sum of NV doubles per lane through LDS, transposed: every lane stores its values, one thread per (row, group)
// adds `chunk` lanes (odd: conflict-free reads), NV threads add the group sums and write partials[row * pstride + bid].
// ~(NV + chunk) instructions per wave where NV DPP wave reductions (wave_sum) take ~23 NV; fixed order -> bitwise
// reproducible.  (k_accum: 18 rows, 19-lane chunks, 14 groups;  point-to-plane rows: 29 / 33 / 8.)
template <int NV>
struct BlockSum {
  static constexpr int kG0 = (int)kBlock / NV;
  static constexpr int kChunk = (((int)kBlock + kG0 - 1) / kG0) | 1;
  static constexpr int kGroups = ((int)kBlock + kChunk - 1) / kChunk;
  static_assert(NV * kGroups <= (int)kBlock, "one thread per (row, group)");
  double tr[NV][kBlock + 1];
  double p1[NV][kGroups];
};

template <int NV>
__device__ __forceinline__ void block_sum_rows_raw(const double* v, double (*tr)[kBlock + 1], double* p1,
                                                   double* __restrict__ partials, uint32_t pstride, uint32_t bid,
                                                   bool has_values = true) {  // (false: a lane beyond kBlock of a wider workgroup)
  constexpr int G = BlockSum<NV>::kGroups, C = BlockSum<NV>::kChunk;
  if (has_values) {
#pragma unroll
    for (int j = 0; j < NV; j++) tr[j][threadIdx.x] = v[j];
  }
  __syncthreads();
  if (threadIdx.x < NV * G) {
    const int j = threadIdx.x / G, g = threadIdx.x % G;
    const int l0 = g * C;
    double sum = tr[j][l0];
#pragma unroll
    for (int i = 1; i < C; i++)
      if (l0 + i < (int)kBlock) sum += tr[j][l0 + i];
    p1[j * G + g] = sum;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double sum = p1[threadIdx.x * G];
#pragma unroll
    for (int g = 1; g < G; g++) sum += p1[threadIdx.x * G + g];
    ((double MH_AS_GLOBAL*)partials)[threadIdx.x * pstride + bid] = sum;  // (global space spelled out: mh_nn_device.h)
  }
}

template <int NV>
__device__ __forceinline__ void block_sum_rows(const double* v, BlockSum<NV>& sh, double* __restrict__ partials,
                                               uint32_t pstride, uint32_t bid) {
  block_sum_rows_raw<NV>(v, sh.tr, &sh.p1[0][0], partials, pstride, bid);
}

// The same with the four lanes of every DPP quad added first (two quad_perm steps per value, VALU only): a quarter of the
// LDS (k_accum: 9.6 KiB per workgroup instead of 37.6, which had capped it at four waves per SIMD) and a quarter of the
// transposed reads.  Fixed order as well.
template <int NV>
struct BlockSumQ {
  static constexpr int kL = (int)kBlock / 4;
  static constexpr int kG0 = kL / NV;
  static constexpr int kChunk = ((kL + kG0 - 1) / kG0) | 1;
  static constexpr int kGroups = (kL + kChunk - 1) / kChunk;
  static_assert(kG0 >= 1 && NV * kGroups <= (int)kBlock, "one thread per (row, group)");
  double tr[NV][kL + 1];
  double p1[NV][kGroups];
};
template <int NV>
__device__ __forceinline__ void block_sum_rows_quad(const double* v, BlockSumQ<NV>& sh, double* __restrict__ partials,
                                                    uint32_t pstride, uint32_t bid) {
  constexpr int G = BlockSumQ<NV>::kGroups, C = BlockSumQ<NV>::kChunk, L = BlockSumQ<NV>::kL;
#pragma unroll
  for (int j = 0; j < NV; j++) {
    double q = v[j];
    q += dpp_f64<0xB1>(q);  // quad_perm:[1,0,3,2]
    q += dpp_f64<0x4E>(q);  // quad_perm:[2,3,0,1]
    if ((threadIdx.x & 3u) == 0u) sh.tr[j][threadIdx.x >> 2] = q;
  }
  __syncthreads();
  if (threadIdx.x < NV * G) {
    const int j = threadIdx.x / G, g = threadIdx.x % G;
    const int l0 = g * C;
    double sum = sh.tr[j][l0];
#pragma unroll
    for (int i = 1; i < C; i++)
      if (l0 + i < L) sum += sh.tr[j][l0 + i];
    sh.p1[j][g] = sum;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double sum = sh.p1[threadIdx.x][0];
#pragma unroll
    for (int g = 1; g < G; g++) sum += sh.p1[threadIdx.x][g];
    ((double MH_AS_GLOBAL*)partials)[threadIdx.x * pstride + bid] = sum;
  }
}

// ================================================================================================
// k_match: correspondence search (+ first Gauss-Newton accumulation when FUSED)
// ================================================================================================
template <bool FUSED, int MODE /* 0: literal 27-voxel scan | 1: exact branch-and-bound */>
__global__ __launch_bounds__(kBlock, MH_MATCH_WAVES) void k_match(const IcpDeviceState* __restrict__ st, PoseArg Targ, float thr2_arg,
                                                  uint32_t apply_thr, const MatchK* __restrict__ kp, const float* __restrict__ lx,
                                                  const float* __restrict__ ly, const float* __restrict__ lz, uint32_t n,
                                                  MapView map, float4* __restrict__ pair_q,
                                                  uint32_t* __restrict__ pair_gidx, double* __restrict__ partials,
                                                  uint32_t pstride) {
  __shared__ BlockSum<kAccN> lds;
  const MatchK k = *kp;  // wave-uniform scalar loads
  double T[12];
  float thr2;
  double kparam = 0.0;
  if (FUSED) {
    if (st->done) return;  // wave-uniform
    const uint32_t it = st->iter;
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = st->T[i];
    const double thr = k.thr[it];
    thr2 = (float)(thr * thr);
    kparam = k.kparam[it];
  } else {
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = Targ.m[i];
    thr2 = thr2_arg;
  }
  const uint32_t bid = blockIdx.x;
  const uint32_t i = bid * kBlock + threadIdx.x;
  Acc a;
  acc_zero(a);
  if (i < n) {
    const float x = lx[i], y = ly[i], z = lz[i];
    float px, py, pz;
    transform_point(T, x, y, z, px, py, pz);
    const NNResult r = MODE == 1 ? nn_search_pruned(map, px, py, pz) : nn_single_search(map, px, py, pz);
    bool ok = r.found;
    if (FUSED || apply_thr) {
      const float n2 = (px * px + py * py) + pz * pz;
      ok = ok && (r.d2 < thr2 + k.ang2 * n2);
    }
    pair_q[i] = make_float4(r.pt.x, r.pt.y, r.pt.z, r.d2);
    pair_gidx[i] = ok ? __float_as_uint(r.pt.w) : kNoMatch;
    if (FUSED && ok) acc_pt2pt(a, T, x, y, z, r.pt.x, r.pt.y, r.pt.z, k.kernel, kparam, k.w_pt2pt);
  }
  if (FUSED) block_sum_rows<kAccN>(a.v, lds, partials, pstride, bid);
}


// Matcher_Points_DistanceThreshold with pairingsPerPoint = k > 1 (rgbd.yaml:135-141): entry i*k + r of the pair buffers is
// the r-th nearest neighbour of point i, valid while the distances pass the threshold ("break at first failure": the limit is
// the same for all of a point's neighbours and they come in ascending distance, so the passing ones are a prefix)
__global__ __launch_bounds__(kBlock) void k_match_kbest(PoseArg Targ, float thr2, float ang2, uint32_t k, const float* __restrict__ lx,
                                                        const float* __restrict__ ly, const float* __restrict__ lz, uint32_t n,
                                                        MapView map, float4* __restrict__ pair_q, uint32_t* __restrict__ pair_gidx) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  double T[12];
#pragma unroll
  for (int j = 0; j < 12; j++) T[j] = Targ.m[j];
  float px, py, pz;
  transform_point(T, lx[i], ly[i], lz[i], px, py, pz);
  knnkey_t best[kMaxKnn];
  nn_search_kbest(map, px, py, pz, k, best);
  const float n2 = (px * px + py * py) + pz * pz;
  const float lim = thr2 + ang2 * n2;
  const gpts_ptr pts4 = (gpts_ptr)map.pts;
  for (uint32_t r = 0; r < k; r++) {
    const knnkey_t key = knn_select(best, r);
    const bool found = key != ~0ull;
    const float d2 = __uint_as_float((uint32_t)(key >> 32));
    f32x4 pt = (f32x4)(0.f);
    if (found) pt = pts4[(uint32_t)key];
    pair_q[(size_t)i * k + r] = make_float4(pt.x, pt.y, pt.z, d2);
    pair_gidx[(size_t)i * k + r] = (found && d2 < lim) ? __float_as_uint(pt.w) : kNoMatch;
  }
}
// Matcher_Point2Plane on a plain point map (pipelines/rgbd.yaml:143-151; SURVEY 8a row a13 "otherwise KNN + PCA" [U]): one lane
// per point -- the knn nearest records of the 27-voxel block (nn_search_kbest), the prefix of them inside the search radius,
// mean + covariance in fp64, cyclic Jacobi (the operation sequence of k_ndt_stats and of the oracle), plane test e0 <= thr * e2,
// distance test in fp64.  pl_c = {centroid, 1 | 0}, pl_n = {unit normal (largest component positive), 0}: what the
// point-to-plane rows and compact_pl_pairs read.  Not on a target pipeline's path: exactness first.
struct PlKnnArg {
  double distance_threshold, plane_eigen_threshold;
  float radius2;
  uint32_t knn, min_pts;
};
__global__ __launch_bounds__(kBlock) void k_match_pl_knn(PoseArg Targ, PlKnnArg a, const float* __restrict__ lx, const float* __restrict__ ly,
                                                         const float* __restrict__ lz, uint32_t n, MapView map, float4* __restrict__ pl_c,
                                                         float4* __restrict__ pl_n) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  double T[12];
#pragma unroll
  for (int j = 0; j < 12; j++) T[j] = Targ.m[j];
  float px, py, pz;
  transform_point(T, lx[i], ly[i], lz[i], px, py, pz);
  float4 rc = make_float4(0.f, 0.f, 0.f, 0.f), rn = make_float4(0.f, 0.f, 0.f, 0.f);
  knnkey_t best[kMaxPlaneKnn];
  nn_search_kbest(map, px, py, pz, a.knn, best);
  // ascending distances: the neighbours inside the radius are a prefix of the list
  uint32_t cnt = 0;
#pragma unroll
  for (int r = 0; r < kMaxPlaneKnn; r++) {
    const bool in = (uint32_t)r < a.knn && best[r] != ~0ull && __uint_as_float((uint32_t)(best[r] >> 32)) < a.radius2;
    cnt += (in && cnt == (uint32_t)r) ? 1u : 0u;
  }
  if (cnt >= a.min_pts) {
    const gpts_ptr pts4 = (gpts_ptr)map.pts;
    f32x4 nb[kMaxPlaneKnn];
#pragma unroll
    for (int r = 0; r < kMaxPlaneKnn; r++) nb[r] = pts4[(uint32_t)r < cnt ? (uint32_t)best[r] : (uint32_t)best[0]];
    double mu[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < kMaxPlaneKnn; r++)
      if ((uint32_t)r < cnt) { mu[0] += (double)nb[r].x; mu[1] += (double)nb[r].y; mu[2] += (double)nb[r].z; }
    mu[0] /= (double)cnt; mu[1] /= (double)cnt; mu[2] /= (double)cnt;
    double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
#pragma unroll
    for (int r = 0; r < kMaxPlaneKnn; r++)
      if ((uint32_t)r < cnt) {
        const double d0 = (double)nb[r].x - mu[0], d1 = (double)nb[r].y - mu[1], d2 = (double)nb[r].z - mu[2];
        c00 += d0 * d0; c01 += d0 * d1; c02 += d0 * d2; c11 += d1 * d1; c12 += d1 * d2; c22 += d2 * d2;
      }
    const double inv = (double)(cnt - 1);
    double A[3][3] = {{c00 / inv, c01 / inv, c02 / inv}, {c01 / inv, c11 / inv, c12 / inv}, {c02 / inv, c12 / inv, c22 / inv}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 12; sweep++) {
#pragma unroll
      for (int pq = 0; pq < 3; pq++) {
        const int p = (pq == 2) ? 1 : 0, q = (pq == 0) ? 1 : 2, r = 3 - p - q;
        const double apq = A[p][q];
        if (apq != 0.0) {
          const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
          const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
          const double app = A[p][p] - t * apq, aqq = A[q][q] + t * apq;
          const double arp = c * A[r][p] - sn * A[r][q], arq = sn * A[r][p] + c * A[r][q];
          A[p][p] = app; A[q][q] = aqq; A[p][q] = 0.0; A[q][p] = 0.0;
          A[r][p] = arp; A[p][r] = arp; A[r][q] = arq; A[q][r] = arq;
#pragma unroll
          for (int v = 0; v < 3; v++) {
            const double vip = c * V[v][p] - sn * V[v][q], viq = sn * V[v][p] + c * V[v][q];
            V[v][p] = vip; V[v][q] = viq;
          }
        }
      }
    }
    // smallest / largest eigenvalue, the eigenvector of the smallest (first minimum: the oracle's stable sort)
    const double w0 = A[0][0], w1 = A[1][1], w2 = A[2][2];
    int imin = 0;
    double wmin = w0, wmax = w0;
    if (w1 < wmin) { wmin = w1; imin = 1; }
    if (w2 < wmin) { wmin = w2; imin = 2; }
    if (w1 > wmax) wmax = w1;
    if (w2 > wmax) wmax = w2;
    if (wmax > 0.0 && !(wmin > a.plane_eigen_threshold * wmax)) {
      double nv[3] = {imin == 0 ? V[0][0] : (imin == 1 ? V[0][1] : V[0][2]), imin == 0 ? V[1][0] : (imin == 1 ? V[1][1] : V[1][2]),
                      imin == 0 ? V[2][0] : (imin == 1 ? V[2][1] : V[2][2])};
      const double len = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
      int big = 0;
      if (fabs(nv[1]) > fabs(nv[big])) big = 1;
      if (fabs(nv[2]) > fabs(nv[big])) big = 2;
      const double sgn = ((big == 0 ? nv[0] : (big == 1 ? nv[1] : nv[2])) < 0.0 ? -1.0 : 1.0) / len;
      nv[0] *= sgn; nv[1] *= sgn; nv[2] *= sgn;
      const double dist = fabs((nv[0] * ((double)px - mu[0]) + nv[1] * ((double)py - mu[1])) + nv[2] * ((double)pz - mu[2]));
      if (!(dist > a.distance_threshold)) {
        rc = make_float4((float)mu[0], (float)mu[1], (float)mu[2], 1.f);
        rn = make_float4((float)nv[0], (float)nv[1], (float)nv[2], 0.f);
      }
    }
  }
  pl_c[i] = rc;
  pl_n[i] = rn;
}
__global__ void k_div_idx(uint32_t* __restrict__ idx, uint32_t n, uint32_t k) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] /= k;
}

#ifdef MH_DEBUG_WAVETRACE
// debug build only: wall_clock64 (100 MHz) at numbered points of the one-workgroup kernels, last launch wins
__device__ unsigned long long g_phase[32];
#define MH_PHASE(i) do { if (threadIdx.x == 0) g_phase[i] = wall_clock64(); } while (0)
extern "C" __attribute__((visibility("default"))) int mh_debug_phases(unsigned long long* host_out) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase), sizeof(g_phase)) == hipSuccess ? 0 : 2;
}
extern "C" __attribute__((visibility("default"))) int mh_debug_flat_counters(unsigned long long* host_out, int reset) {
  (void)hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mh::g_flatdbg), sizeof(mh::g_flatdbg)) != hipSuccess) return 2;
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(mh::g_flatdbg), z, sizeof(z)); }
  return 0;
}
// k_icp16 (a loop): the time between consecutive stamps is ACCUMULATED per phase, workgroup 0's first lane, written out at the end
#define MH_LOOP_STAMPS unsigned long long lp_t = wall_clock64(), lp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define MH_LOOP_STAMP(i) do { const unsigned long long now_ = wall_clock64(); lp_acc[i] += now_ - lp_t; lp_t = now_; } while (0)
#define MH_LOOP_STAMPS_OUT(steps) do { if (blockIdx.x == 0 && threadIdx.x == 0) { for (int q_ = 0; q_ < 12; q_++) g_phase[16 + q_] = lp_acc[q_]; g_phase[28] = (steps); } } while (0)
#else
#define MH_PHASE(i) do { } while (0)
#define MH_LOOP_STAMPS do { } while (0)
#define MH_LOOP_STAMP(i) do { } while (0)
#define MH_LOOP_STAMPS_OUT(steps) do { } while (0)
#endif
#ifdef MH_DEBUG_WAVETRACE
static unsigned long long* g_wtrace = nullptr;  // debug build only: [2 * n_waves] begin/end wall_clock64 of the last launch
extern "C" __attribute__((visibility("default"))) int mh_debug_wavetrace(unsigned long long* host_out, size_t n_waves) {
  if (!g_wtrace) { if (hipMalloc(&g_wtrace, 16u << 20) != hipSuccess) return 1; (void)hipMemset(g_wtrace, 0, 16u << 20); return 0; }
  (void)hipDeviceSynchronize();
  return hipMemcpy(host_out, g_wtrace, n_waves * 16, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}
#endif
// ================================================================================================
// k_match4: correspondence search with a DPP quad per scan point (nn_search_quad).  Device-state driven like the
// fused k_match, but it only stores the pairings: the first Gauss-Newton accumulation is the k_accum launch that
// follows (64 points per wave there, 16 here).
// ================================================================================================
__device__ __forceinline__ void k_match4_body(const IcpDeviceState* __restrict__ st,
                                                   const float* __restrict__ lx, const float* __restrict__ ly,
                                                   const float* __restrict__ lz, uint32_t n, MapView map,
                                                   float4* __restrict__ pair_q, uint32_t* __restrict__ pair_gidx,
                                                   const uint32_t* __restrict__ perm  // null, or lx/ly/lz are the scan in
                                                                                      // search order: point i is perm[i]
#ifdef MH_DEBUG_WAVETRACE
                                                   , unsigned long long* __restrict__ wtrace
#endif
) {
#ifdef MH_DEBUG_WAVETRACE
  struct WT { unsigned long long* p; unsigned long long t0; uint32_t w;
              __device__ ~WT() { if ((threadIdx.x & 63) == 0 && p) { p[2 * w] = t0; p[2 * w + 1] = wall_clock64(); } } }
      wt{wtrace, (unsigned long long)wall_clock64(), (blockIdx.x * kBlock + threadIdx.x) >> 6};
#endif
  // everything needed per iteration sits in the state block (k_solve publishes the next threshold there): one batch of
  // scalar loads instead of the chain state -> parameter block -> threshold table, and the point is fetched alongside
  const uint32_t gl = blockIdx.x * kBlock + threadIdx.x;
  const uint32_t i = gl >> 2, sub = gl & 3u;
  const uint32_t ic = i < n ? i : n - 1;
  const uint32_t o = perm ? G(perm)[ic] : ic;  // (clamped: lanes past the end of a short job of a batch read, and never write)
  const float x = G(lx)[ic], y = G(ly)[ic], z = G(lz)[ic];
  // the state block through the scalar path (uniform address, not written during this kernel): the pose in SGPRs
  typedef const IcpDeviceState __attribute__((address_space(4))) * cstate_ptr;
  const cstate_ptr cst = (cstate_ptr)uniform_const_ptr(st);
  const uint32_t done = cst->done;
  // From the second ICP iteration on, pair_q[o] still holds the record this point was paired with under the previous
  // pose: its distance under the new pose bounds the search (nn_search_quad).  Iteration 0 of every alignment starts
  // without one (the buffer may hold another scan's pairings).
  const bool have_prev = cst->iter > 0 && !map.no_prev_bound;
  f32x4 prev = (f32x4){0.f, 0.f, 0.f, __builtin_inff()};
  if (have_prev) prev = G(reinterpret_cast<const f32x4*>(pair_q))[o];  // grid-uniform branch
  double T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = cst->T[k];
  const float thr2 = cst->cur_thr2, ang2 = cst->cur_ang2;
  if (done) return;  // wave-uniform
  if (i >= n) return;  // whole quads
  float px, py, pz;
  transform_point(T, x, y, z, px, py, pz);
  float bound0 = __builtin_inff();
  if (prev.w < __builtin_inff()) {  // a record was found last time (whatever the threshold said)
    const float dx = prev.x - px, dy = prev.y - py, dz = prev.z - pz;
    bound0 = (dx * dx + dy * dy) + dz * dz;  // the candidate arithmetic of nn_scan_round_quad
  }
  const NNResult r = nn_search_quad(map, sub, px, py, pz, bound0);
  if (sub == 0) {
    const float n2 = (px * px + py * py) + pz * pz;
    const bool ok = r.found && (r.d2 < thr2 + ang2 * n2);
    G(reinterpret_cast<f32x4*>(pair_q))[o] = (f32x4){r.pt.x, r.pt.y, r.pt.z, r.d2};
    G(pair_gidx)[o] = ok ? __float_as_uint(r.pt.w) : kNoMatch;
  }
}

// k_match16 (below, after the point-to-plane row search it can carry along): the same step with a DPP row (16 lanes)
// per scan point (nn_search_row16): for small layers, where the launch is pure latency; chosen automatically below
// kRowMaxPoints points.
constexpr uint32_t kFused16MaxPoints = 12288;  // up to here the row kernel also accumulates the first Gauss-Newton step
constexpr uint32_t kRowMaxPoints = 32768;  // measured cross-over with the quad kernel: ~40 k points (C2 map)
// ================================================================================================
// k_accum: point-to-point accumulation on stored pairings (inner GN steps, solver-granular path)
// ================================================================================================
#ifndef MH_ACC_PPT
#define MH_ACC_PPT 4
#endif
constexpr uint32_t kAccPPT = MH_ACC_PPT;  // scan points per lane of k_accum (tools/build_variants.sh: 2 and 8 measured)
inline uint32_t nblk_acc(size_t n) { return (uint32_t)((n + (size_t)kBlock * kAccPPT - 1) / ((size_t)kBlock * kAccPPT)); }

// SIGNED: the verdict rides in the sign of the pairing's distance (the plan / scan matcher, flat_signed_d2): no index array read
template <bool SIGNED>
__device__ __forceinline__ void k_accum_body(const IcpDeviceState* __restrict__ st, uint32_t first,
                                                  const MatchK* __restrict__ kp, const float* __restrict__ lx,
                                                  const float* __restrict__ ly, const float* __restrict__ lz, uint32_t n,
                                                  const float4* __restrict__ pair_q,
                                                  const uint32_t* __restrict__ pair_gidx, double* __restrict__ partials,
                                                  uint32_t pstride, uint32_t block_x) {
  __shared__ BlockSumQ<kAccN> bs;
  // state and parameters through the scalar path (uniform addresses, not written during this kernel); the arrays through
  // global-space pointers (mh_nn_device.h, G())
  typedef const IcpDeviceState __attribute__((address_space(4))) * cstate_ptr;
  typedef const MatchK __attribute__((address_space(4))) * cmatchk_ptr;
  const cstate_ptr cst = (cstate_ptr)uniform_const_ptr(st);
  const cmatchk_ptr ck = (cmatchk_ptr)uniform_const_ptr(kp);
  if (cst->done) return;
  if (!first && cst->inner == 0) return;  // the previous solve already closed this ICP iteration
  double T[12];
#pragma unroll
  for (int i = 0; i < 12; i++) T[i] = cst->T[i];
  struct { uint32_t kernel; double w_pt2pt; } k = {ck->kernel, ck->w_pt2pt};
  const double kparam = cst->cur_kparam;
  // kAccPPT points per lane: the reduction below is a fixed cost per lane, amortised over four points
  // (the device is VALU-bound once several alignments run concurrently)
  const uint32_t bid = block_x;
  uint32_t gi[kAccPPT];
  f32x4 q[kAccPPT];
  float px[kAccPPT], py[kAccPPT], pz[kAccPPT];
  const auto gq = G(reinterpret_cast<const f32x4*>(pair_q));
#pragma unroll
  for (int u = 0; u < kAccPPT; u++) {  // all loads first (clamped index), then the arithmetic
    const uint32_t i = (bid * kAccPPT + (uint32_t)u) * kBlock + threadIdx.x;
    const uint32_t ic = i < n ? i : n - 1;
    q[u] = gq[ic];
    if (SIGNED) gi[u] = (i < n && !(__float_as_uint(q[u].w) >> 31)) ? 0u : kNoMatch;
    else gi[u] = i < n ? G(pair_gidx)[ic] : kNoMatch;
    px[u] = G(lx)[ic]; py[u] = G(ly)[ic]; pz[u] = G(lz)[ic];
  }
  Acc a;
  acc_zero(a);
#pragma unroll
  for (int u = 0; u < kAccPPT; u++)
    acc_pt2pt_masked(a, T, gi[u] != kNoMatch, px[u], py[u], pz[u], q[u].x, q[u].y, q[u].z, k.kernel, kparam, k.w_pt2pt);
  block_sum_rows_quad<kAccN>(a.v, bs, partials, pstride, bid);
}

// point-to-plane rows (Matcher_Point2Plane pairings, lidar3d-ndt.yaml:195-200): e = n.(R l + t - c),
// J = [ (R^T n)^T | (l x R^T n)^T ].  Generic partial: 21 upper-triangle H + 6 g + cost + count.
constexpr int kGenN = 29;
__global__ __launch_bounds__(kBlock) void k_accum_pl(const IcpDeviceState* __restrict__ st, uint32_t kernel,
                                                     double kparam, double wpair, const float* __restrict__ l3,
                                                     const float* __restrict__ c3, const float* __restrict__ n3,
                                                     uint32_t n, uint32_t stride, double* __restrict__ partials,
                                                     uint32_t pstride) {
  __shared__ BlockSum<kGenN> lds;
  if (st->done) return;
  double T[12];
#pragma unroll
  for (int i = 0; i < 12; i++) T[i] = st->T[i];
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  double v[kGenN];
#pragma unroll
  for (int j = 0; j < kGenN; j++) v[j] = 0.0;
  if (i < n) {
    const double lx = l3[i], ly = l3[stride + i], lz = l3[2 * stride + i];
    const double cx = c3[i], cy = c3[stride + i], cz = c3[2 * stride + i];
    const double nx = n3[i], ny = n3[stride + i], nz = n3[2 * stride + i];
    const double gx = T[0] * lx + T[1] * ly + T[2] * lz + T[3] - cx;
    const double gy = T[4] * lx + T[5] * ly + T[6] * lz + T[7] - cy;
    const double gz = T[8] * lx + T[9] * ly + T[10] * lz + T[11] - cz;
    const double e = nx * gx + ny * gy + nz * gz;
    const double w = wpair * robust_weight(kernel, kparam, e * e);
    double J[6];
    J[0] = T[0] * nx + T[4] * ny + T[8] * nz;  // m = R^T n
    J[1] = T[1] * nx + T[5] * ny + T[9] * nz;
    J[2] = T[2] * nx + T[6] * ny + T[10] * nz;
    J[3] = ly * J[2] - lz * J[1];
    J[4] = lz * J[0] - lx * J[2];
    J[5] = lx * J[1] - ly * J[0];
    int q = 0;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int b = a; b < 6; b++) v[q++] = w * J[a] * J[b];
#pragma unroll
    for (int a = 0; a < 6; a++) v[21 + a] = w * J[a] * e;
    v[27] = w * e * e;
    v[28] = 1.0;
  }
  block_sum_rows<kGenN>(v, lds, partials, pstride, blockIdx.x);
}

// ================================================================================================
// k_solve: one wave.  Ordered reduction of the block partials, prior factor, LDL^T solve, SE(3)
// retraction, inner/outer loop bookkeeping (optimal_tf_gauss_newton + the tail of ICP::align's loop).
// ================================================================================================
// ================================================================================================
// Matcher_Point2Plane on an NDT map (lidar3d-ndt.yaml:195-200): nearest planar voxel of the 27-block
// by centroid distance, accepted iff |n.(p'-c)| < threshold.  One lane per scan point; the 27 slot probes go out in
// three batches of nine unconditional loads, the nine centroid records of a batch likewise.
// ================================================================================================
__device__ __forceinline__ void acc_pt2pl_rows(double* v, const double* __restrict__ T, float lxf, float lyf, float lzf,
                                               const float4& c, const float4& nrm, uint32_t kernel, double kparam,
                                               double wpair) {
  const double lx = lxf, ly = lyf, lz = lzf;
  const double gx = T[0] * lx + T[1] * ly + T[2] * lz + T[3] - (double)c.x;
  const double gy = T[4] * lx + T[5] * ly + T[6] * lz + T[7] - (double)c.y;
  const double gz = T[8] * lx + T[9] * ly + T[10] * lz + T[11] - (double)c.z;
  const double nx = nrm.x, ny = nrm.y, nz = nrm.z;
  const double e = nx * gx + ny * gy + nz * gz;
  const double w = wpair * robust_weight(kernel, kparam, e * e);
  double J[6];
  J[0] = T[0] * nx + T[4] * ny + T[8] * nz;  // m = R^T n
  J[1] = T[1] * nx + T[5] * ny + T[9] * nz;
  J[2] = T[2] * nx + T[6] * ny + T[10] * nz;
  J[3] = ly * J[2] - lz * J[1];
  J[4] = lz * J[0] - lx * J[2];
  J[5] = lx * J[1] - ly * J[0];
  int q = 0;
#pragma unroll
  for (int a = 0; a < 6; a++)
#pragma unroll
    for (int b = a; b < 6; b++) v[q++] = w * J[a] * J[b];
#pragma unroll
  for (int a = 0; a < 6; a++) v[21 + a] = w * J[a] * e;
  v[27] = w * e * e;
  v[28] = 1.0;
}

// Matcher_Point2Plane's acceptance test.  thr > 0: point-to-plane distance |n.(p'-c)| < thr (default,
// MH_PT2PL_PLANE_DISTANCE); thr < 0 encodes MH_PT2PL_CENTROID_DISTANCE: |p'-c|^2 < thr^2, fp32, un-fused like the search.
__device__ __forceinline__ bool pl_accept(const f32x4& bn, float dx, float dy, float dz, float thr) {
  if (thr < 0.f) return (dx * dx + dy * dy) + dz * dz < thr * thr;
  return fabsf((bn.x * dx + bn.y * dy) + bn.z * dz) < thr;
}

template <bool FUSED>
__global__ __launch_bounds__(kBlock) void k_match_pl(const IcpDeviceState* __restrict__ st, PoseArg Targ, float thr_arg,
                                                     const MatchK* __restrict__ kp, const float* __restrict__ lx,
                                                     const float* __restrict__ ly, const float* __restrict__ lz, uint32_t n,
                                                     MapView map, float4* __restrict__ pl_c, float4* __restrict__ pl_n,
                                                     double* __restrict__ partials, uint32_t pstride) {
  __shared__ BlockSum<kGenN> lds;
  const MatchK k = *kp;
  double T[12];
  float thr;
  double kparam = 0.0;
  if (FUSED) {
    if (st->done) return;
    const uint32_t it = st->iter;
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = st->T[i];
    thr = (float)k.pl_thr[it];
    kparam = k.kparam[it];
  } else {
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = Targ.m[i];
    thr = thr_arg;
  }
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  double v[kGenN];
#pragma unroll
  for (int j = 0; j < kGenN; j++) v[j] = 0.0;
  if (i < n) {
    const float x = lx[i], y = ly[i], z = lz[i];
    float px, py, pz;
    transform_point(T, x, y, z, px, py, pz);
    const float lim = 1.0e6f;
    const bool valid = isfinite(px) && isfinite(py) && isfinite(pz) && fabsf(px * map.inv_vs) < lim &&
                       fabsf(py * map.inv_vs) < lim && fabsf(pz * map.inv_vs) < lim;
    float best = __builtin_inff();
    uint32_t best_first = 0;
    f32x4 bc = (f32x4)(0.f);
    if (valid) {
      const gslots_ptr slots4 = (gslots_ptr)map.slots;
      const gpts_ptr pts4 = (gpts_ptr)map.pts;
      const unsigned long long kbase = pack_key(voxel_of(px, map.inv_vs, map.trunc) - 1, voxel_of(py, map.inv_vs, map.trunc) - 1,
                                                voxel_of(pz, map.inv_vs, map.trunc) - 1);
#pragma unroll 1
      for (int ix = 0; ix < 3; ix++) {  // x outer: scan order is preserved for the first-minimum rule
        u32x4 sl[9];
        uint32_t first[9];
#pragma unroll
        for (int c = 0; c < 9; c++) {
          const unsigned long long key = kbase + ((unsigned long long)ix << 42) + ((unsigned long long)(c / 3) << 21) + (unsigned long long)(c % 3);
          sl[c] = slots4[hash_key(key) & map.mask];
        }
#pragma unroll
        for (int c = 0; c < 9; c++) {
          const unsigned long long key = kbase + ((unsigned long long)ix << 42) + ((unsigned long long)(c / 3) << 21) + (unsigned long long)(c % 3);
          unsigned long long sk = ((unsigned long long)sl[c].y << 32) | sl[c].x;
          if (sk != key && sk != kEmptyKey) {
            uint32_t h = hash_key(key) & map.mask;
            do {
              h = (h + 1) & map.mask;
              sl[c] = slots4[h];
              sk = ((unsigned long long)sl[c].y << 32) | sl[c].x;
            } while (sk != key && sk != kEmptyKey);
          }
          first[c] = (sk == key) ? sl[c].z : 0u;  // 0 = absent (a present voxel has first >= 2)
        }
        f32x4 cen[9];
#pragma unroll
        for (int c = 0; c < 9; c++) cen[c] = pts4[first[c] >= 2u ? first[c] - 2u : 0u];  // unconditional, clamped
#pragma unroll
        for (int c = 0; c < 9; c++)
          if (first[c] >= 2u && cen[c].w != 0.f) {
            const float dx = cen[c].x - px, dy = cen[c].y - py, dz = cen[c].z - pz;
            const float d2 = (dx * dx + dy * dy) + dz * dz;
            if (d2 < best) { best = d2; best_first = first[c]; bc = cen[c]; }
          }
      }
    }
    bool ok = false;
    f32x4 bn = (f32x4)(0.f);
    if (best_first >= 2u) {
      bn = ((gpts_ptr)map.pts)[best_first - 1u];
      const float dx = px - bc.x, dy = py - bc.y, dz = pz - bc.z;
      ok = pl_accept(bn, dx, dy, dz, thr);
    }
    const float4 c4 = make_float4(bc.x, bc.y, bc.z, ok ? 1.f : 0.f), n4 = make_float4(bn.x, bn.y, bn.z, 0.f);
    pl_c[i] = c4;
    pl_n[i] = n4;
    if (FUSED && ok) acc_pt2pl_rows(v, T, x, y, z, c4, n4, k.kernel, kparam, k.w_pt2pl);
  }
  if (FUSED) block_sum_rows<kGenN>(v, lds, partials, pstride, blockIdx.x);
}

// Matcher_Point2Plane with a DPP row (16 lanes) per point, for small layers.  k_match_pl walks the 27 voxels in three
// dependent groups of probes + centroid loads (57 us per launch on a 1 k-point layer); here lane r probes codes r and
// r + 16, reads the two statistics records of its voxels, and the row takes the minimum of (d2 to the centroid, code) --
// code order IS the reference's scan order -- in two round trips: nearest planar voxel of the 27-block by centroid
// distance (first in code order among equals), accepted iff |n.(p'-c)| < thr; every lane of the row returns the same
// centroid / normal / verdict.  Runs inside k_match16<true>.
__device__ __forceinline__ bool pl_row_search(const MapView& map, uint32_t r16, float px, float py, float pz, float thr,
                                              f32x4& bc, f32x4& bn) {
  const float lim = 1.0e6f;
  const bool valid = isfinite(px) && isfinite(py) && isfinite(pz) && fabsf(px * map.inv_vs) < lim &&
                     fabsf(py * map.inv_vs) < lim && fabsf(pz * map.inv_vs) < lim;
  nnkey_t best = kNNKeyNone;  // (d2 bits << 32 | code): first strict minimum in scan order
  f32x4 ca = (f32x4)(0.f), na = (f32x4)(0.f), cb = (f32x4)(0.f), nb = (f32x4)(0.f);
  if (valid) {  // row-uniform
    const gslots_ptr slots4 = (gslots_ptr)map.slots;
    const gpts_ptr pts4 = (gpts_ptr)map.pts;
    const unsigned long long kbase = pack_key(voxel_of(px, map.inv_vs, map.trunc) - 1, voxel_of(py, map.inv_vs, map.trunc) - 1,
                                              voxel_of(pz, map.inv_vs, map.trunc) - 1);
    const int code_a = (int)r16, code_b = (int)r16 + 16;
    const bool has_b = code_b < 27;
    const unsigned long long ka = nn_key_of(kbase, code_a), kb = nn_key_of(kbase, has_b ? code_b : code_a);
    const u32x4 sa = slots4[hash_key(ka) & map.mask];
    const u32x4 sb = slots4[hash_key(kb) & map.mask];
    uint32_t fa, cnt_a, fb, cnt_b;
    nn_resolve(map, slots4, ka, sa, true, fa, cnt_a);
    nn_resolve(map, slots4, kb, sb, has_b, fb, cnt_b);
    const bool pa = cnt_a > 0 || fa >= 2u, pb = has_b && (cnt_b > 0 || fb >= 2u);  // a present voxel has first >= 2
    // both statistics records of both voxels in one round trip (clamped, not predicated)
    ca = pts4[pa ? fa - 2u : 0u];
    na = pts4[pa ? fa - 1u : 0u];
    cb = pts4[pb ? fb - 2u : 0u];
    nb = pts4[pb ? fb - 1u : 0u];
    if (pa && ca.w != 0.f) {
      const float dx = ca.x - px, dy = ca.y - py, dz = ca.z - pz;
      const nnkey_t kk = ((nnkey_t)__float_as_uint((dx * dx + dy * dy) + dz * dz) << 32) | (uint32_t)code_a;
      best = kk < best ? kk : best;
    }
    if (pb && cb.w != 0.f) {
      const float dx = cb.x - px, dy = cb.y - py, dz = cb.z - pz;
      const nnkey_t kk = ((nnkey_t)__float_as_uint((dx * dx + dy * dy) + dz * dz) << 32) | (uint32_t)code_b;
      best = kk < best ? kk : best;
    }
  }
  best = row_min_key(best);
  const uint32_t wcode = nnkey_idx(best);
  bc = (f32x4)(0.f);
  bn = (f32x4)(0.f);
  bool ok = false;
  if (wcode != 0xFFFFFFFFu) {  // row-uniform: the owner lane hands its records to the row
    const bool from_b = wcode >= 16u;
    const uint32_t owner = wcode & 15u;
    const f32x4 mc = from_b ? cb : ca, mn = from_b ? nb : na;
    bc.x = __uint_as_float(row_bcast_u32(__float_as_uint(mc.x), owner));
    bc.y = __uint_as_float(row_bcast_u32(__float_as_uint(mc.y), owner));
    bc.z = __uint_as_float(row_bcast_u32(__float_as_uint(mc.z), owner));
    bn.x = __uint_as_float(row_bcast_u32(__float_as_uint(mn.x), owner));
    bn.y = __uint_as_float(row_bcast_u32(__float_as_uint(mn.y), owner));
    bn.z = __uint_as_float(row_bcast_u32(__float_as_uint(mn.z), owner));
    const float dx = px - bc.x, dy = py - bc.y, dz = pz - bc.z;
    ok = pl_accept(bn, dx, dy, dz, thr);
  }
  return ok;
}

// k_match16: a DPP row (16 lanes) per scan point for layers up to kRowMaxPoints (see nn_search_row16).
// PL: the same launch also runs Matcher_Point2Plane for the point (pl_row_search, pairings into pl_c / pl_n): the NDT
// pipeline's two matchers in one kernel instead of two.
// FUSED: the row leaders also accumulate the first Gauss-Newton step of their pairing and the workgroup writes one
// partial per row of sums (16 points per workgroup): layers of 2-32 k points -- what lidar3d-default.yaml really feeds --
// run match | solve | accumulate | solve, four launches per iteration instead of five.
template <bool PL, bool FUSED>
__device__ __forceinline__ void k_match16_body(const IcpDeviceState* __restrict__ st, const MatchK* __restrict__ kp,
                                                    const float* __restrict__ lx, const float* __restrict__ ly,
                                                    const float* __restrict__ lz, uint32_t n, MapView map,
                                                    float4* __restrict__ pair_q, uint32_t* __restrict__ pair_gidx,
                                                    float4* __restrict__ pl_c, float4* __restrict__ pl_n,
                                                    double* __restrict__ partials, uint32_t pstride) {
  __shared__ double rows[FUSED ? kAccN : 1][kBlock / 16 + 1];
  const uint32_t gl = blockIdx.x * kBlock + threadIdx.x;
  const uint32_t i = gl >> 4, r16 = gl & 15u;
  const uint32_t ic = i < n ? i : n - 1;
  const float x = G(lx)[ic], y = G(ly)[ic], z = G(lz)[ic];
  // state and parameters through the scalar path (uniform addresses, not written during this kernel: mh_nn_device.h)
  typedef const IcpDeviceState __attribute__((address_space(4))) * cstate_ptr;
  typedef const MatchK __attribute__((address_space(4))) * cmatchk_ptr;
  typedef const double __attribute__((address_space(4))) * cf64_ptr;
  const cstate_ptr cst = (cstate_ptr)uniform_const_ptr(st);
  const cmatchk_ptr ck = (cmatchk_ptr)uniform_const_ptr(kp);
  const uint32_t done = cst->done;
  // the record paired with this point under the previous pose bounds the search (k_match4_body has the story)
  const bool have_prev = cst->iter > 0 && !map.no_prev_bound;
  f32x4 prev = (f32x4){0.f, 0.f, 0.f, __builtin_inff()};
  if (have_prev) prev = G(reinterpret_cast<const f32x4*>(pair_q))[ic];  // grid-uniform branch
  double T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = cst->T[k];
  const float thr2 = cst->cur_thr2, ang2 = cst->cur_ang2;
  float pl_thr = 0.f;
  if (PL) pl_thr = (float)((cf64_ptr)uniform_const_ptr(ck->pl_thr))[cst->iter];
  uint32_t kernel = 0;
  double kparam = 0.0, wpair = 0.0;
  if (FUSED) {
    kernel = ck->kernel;
    wpair = ck->w_pt2pt;
    kparam = cst->cur_kparam;
  }
  if (done) return;              // grid-uniform
  if (!FUSED && i >= n) return;  // whole rows (FUSED: they stay for the barrier)
  Acc a;
  acc_zero(a);
  if (i < n) {  // row-uniform
    float px, py, pz;
    transform_point(T, x, y, z, px, py, pz);
    float bound0 = __builtin_inff();
    if (prev.w < __builtin_inff()) {
      const float dx = prev.x - px, dy = prev.y - py, dz = prev.z - pz;
      bound0 = (dx * dx + dy * dy) + dz * dz;  // the candidate arithmetic of the scans
    }
    const NNResult r = nn_search_row16(map, r16, px, py, pz, bound0);
    const float n2 = (px * px + py * py) + pz * pz;
    bool ok = r.found && (r.d2 < thr2 + ang2 * n2);
    if (PL) {  // Matcher_Point2Plane runs first in the reference's order; its verdict may keep the point out of the point matcher
      f32x4 bc, bn;
      const bool okp = pl_row_search(map, r16, px, py, pz, pl_thr, bc, bn);
      if (r16 == 0) {
        G(reinterpret_cast<f32x4*>(pl_c))[i] = (f32x4){bc.x, bc.y, bc.z, okp ? 1.f : 0.f};
        G(reinterpret_cast<f32x4*>(pl_n))[i] = (f32x4){bn.x, bn.y, bn.z, 0.f};
      }
      if (okp && ck->skip_pl_paired) ok = false;  // (the nearest point still goes to pair_q: it bounds the next search)
    }
    if (r16 == 0) {
      G(reinterpret_cast<f32x4*>(pair_q))[i] = (f32x4){r.pt.x, r.pt.y, r.pt.z, r.d2};
      G(pair_gidx)[i] = ok ? __float_as_uint(r.pt.w) : kNoMatch;
    }
    if (FUSED && r16 == 0) acc_pt2pt_masked(a, T, ok, x, y, z, r.pt.x, r.pt.y, r.pt.z, kernel, kparam, wpair);
  }
  if (FUSED) {  // 16 row leaders per workgroup -> one partial per sum, fixed order
    if (r16 == 0) {
#pragma unroll
      for (int j = 0; j < kAccN; j++) rows[j][threadIdx.x >> 4] = a.v[j];
    }
    __syncthreads();
    if (threadIdx.x < kAccN) {
      double sum = rows[threadIdx.x][0];
#pragma unroll
      for (int q = 1; q < (int)(kBlock / 16); q++) sum += rows[threadIdx.x][q];
      G(partials)[threadIdx.x * pstride + blockIdx.x] = sum;
    }
  }
}

// Point-to-point moments AND the Gauss-Newton rows of the stored point-to-plane pairings in one launch (NDT maps,
// layers above the one-workgroup size; `first`: also when the iteration has just begun): four points per lane, both
// kinds of rows, the two workgroup sums share one transposed buffer.  partials / partials_b both get gridDim.x columns.
__global__ __launch_bounds__(kBlock) void k_accum_both(const IcpDeviceState* __restrict__ st, uint32_t first,
                                                       const MatchK* __restrict__ kp, const float* __restrict__ lx,
                                                       const float* __restrict__ ly, const float* __restrict__ lz, uint32_t n,
                                                       const float4* __restrict__ pair_q,
                                                       const uint32_t* __restrict__ pair_gidx,
                                                       const float4* __restrict__ pl_c, const float4* __restrict__ pl_n,
                                                       double* __restrict__ partials, double* __restrict__ partials_b,
                                                       uint32_t pstride) {
  __shared__ double tr[kGenN][kBlock + 1];
  __shared__ double p1[(kAccN * BlockSum<kAccN>::kGroups > kGenN * BlockSum<kGenN>::kGroups) ? kAccN * BlockSum<kAccN>::kGroups
                                                                                               : kGenN * BlockSum<kGenN>::kGroups];
  if (st->done || (!first && st->inner == 0)) return;
  double T[12];
#pragma unroll
  for (int i = 0; i < 12; i++) T[i] = st->T[i];
  const MatchK k = *kp;
  const double kparam = st->cur_kparam;
  const uint32_t bid = blockIdx.x;
  uint32_t gi[kAccPPT];
  float4 q[kAccPPT], pc[kAccPPT], pn[kAccPPT];
  float px[kAccPPT], py[kAccPPT], pz[kAccPPT];
#pragma unroll
  for (int u = 0; u < kAccPPT; u++) {  // all loads first (clamped index), then the arithmetic
    const uint32_t i = (bid * kAccPPT + (uint32_t)u) * kBlock + threadIdx.x;
    const uint32_t ic = i < n ? i : n - 1;
    gi[u] = i < n ? pair_gidx[ic] : kNoMatch;
    q[u] = pair_q[ic];
    px[u] = lx[ic]; py[u] = ly[ic]; pz[u] = lz[ic];
    pc[u] = pl_c[ic];
    pn[u] = pl_n[ic];
    if (i >= n) pc[u].w = 0.f;
  }
  Acc a;
  acc_zero(a);
  double v[kGenN];
#pragma unroll
  for (int j = 0; j < kGenN; j++) v[j] = 0.0;
#pragma unroll
  for (int u = 0; u < kAccPPT; u++) {
    acc_pt2pt_masked(a, T, gi[u] != kNoMatch, px[u], py[u], pz[u], q[u].x, q[u].y, q[u].z, k.kernel, kparam, k.w_pt2pt);
    if (pc[u].w != 0.f) {
      double r[kGenN];
      acc_pt2pl_rows(r, T, px[u], py[u], pz[u], pc[u], pn[u], k.kernel, kparam, k.w_pt2pl);
#pragma unroll
      for (int j = 0; j < kGenN; j++) v[j] += r[j];
    }
  }
  block_sum_rows_raw<kAccN>(a.v, tr, p1, partials, pstride, bid);
  __syncthreads();  // the buffer is reused
  block_sum_rows_raw<kGenN>(v, tr, p1, partials_b, pstride, bid);
}

constexpr int kSolveThreads = 512;  // 2 waves per SIMD -> 256 VGPRs for the serial 6x6 code of thread 0

// Ordered sum of `nvals` rows of a [nvals][stride] array of per-block partials over n blocks, by the
// whole block: G = blockDim/nvals lanes per row, 8 independent loads in flight per lane, then
// a fixed-order LDS pass.  Shape depends only on (n, nvals) -> bitwise reproducible.
__device__ __forceinline__ void reduce_rows(const double* __restrict__ part, uint32_t n, uint32_t stride, int nvals,
                                            double* __restrict__ out, double (*red)[64]) {
  const int t = threadIdx.x;
  int G = kSolveThreads / nvals;
  if (G > 64) G = 64;
  const int v = t / G, g = t % G;
  if (v < nvals) {
    const double MH_AS_GLOBAL* src = (const double MH_AS_GLOBAL*)part + (size_t)v * stride;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0, s6 = 0.0, s7 = 0.0;
    uint32_t b = g;
    for (; b + 7u * G < n; b += 8u * G) {  // 8 independent loads in flight per lane
      const double v0 = src[b], v1 = src[b + G], v2 = src[b + 2u * G], v3 = src[b + 3u * G];
      const double v4 = src[b + 4u * G], v5 = src[b + 5u * G], v6 = src[b + 6u * G], v7 = src[b + 7u * G];
      s0 += v0; s1 += v1; s2 += v2; s3 += v3; s4 += v4; s5 += v5; s6 += v6; s7 += v7;
    }
    for (; b < n; b += 8u * G) {  // the remainder (all of it below 8 G columns): into the first sum, in order -- its loads together
      double w[8];
#pragma unroll
      for (uint32_t u = 0; u < 8u; u++) w[u] = b + u * G < n ? src[b + u * G] : 0.0;
#pragma unroll
      for (uint32_t u = 0; u < 8u; u++)
        if (b + u * G < n) s0 += w[u];
    }
    red[v][g] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
  }
  __syncthreads();
  if (t < nvals) {  // (eight reads at a time ahead of their additions, the additions in order: a read per addition costs its LDS latency G times over)
    double acc = 0.0;
    for (int q0 = 0; q0 < G; q0 += 8) {
      double part[8];
#pragma unroll
      for (int u = 0; u < 8; u++) part[u] = red[t][q0 + u < G ? q0 + u : G - 1];
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (q0 + u < G) acc += part[u];
    }
    out[t] = acc;
  }
  __syncthreads();
}

struct SolveShared {
  double red[kGenN][64];  // reduce_rows' scratch (partials from global memory)
  double totA[kAccN], totB[kGenN];
  double sh_log[13][6];
};
// ... without the scratch, for callers that hand solve_body ready totals (k_icpw keeps its reduction scratch elsewhere)
struct SolveSharedTotals {
  double totA[kAccN], totB[kGenN];
  double sh_log[13][6];
};
__device__ __forceinline__ double (*solve_red(SolveShared& s))[64] { return s.red; }
__device__ __forceinline__ double (*solve_red(SolveSharedTotals&))[64] { return nullptr; }

// One Gauss-Newton step + the tail of the ICP iteration, executed by ONE workgroup of kSolveThreads lanes.  Every lane
// must call it; only lane 0 runs the serial part.  (A cooperative single-launch version of the whole loop for the
// 1-8 k-point layers of the real pipeline -- match | grid barrier | solve | grid barrier | accumulate ... -- was
// built on top of this and measured: 1.91 vs 1.98 ms of ICP per scan, i.e. the launch boundaries are not what a
// small alignment waits for; it was removed again.  So was a "last workgroup of k_accum runs the solve" fusion
// (ticket counter + __threadfence): correct, but the device-scope release/acquire fences write back and invalidate the
// XCDs' L2s on every launch -- the map falls out of cache and C2 drops from 2285 to 960 scans/s.  Kernel boundaries
// are the cheap way to order producers and consumers on this part.)
// LDS_STATE: the state block lives in LDS (k_step16: one copy per workgroup) instead of global memory.
// WAVE0: only the first wave of the workgroup calls (the totals are ready in LDS, nothing here needs the other waves): the
// one workgroup barrier below becomes a wave-level hand-over.
template <bool LDS_STATE = false, bool WAVE0 = false, class SH = SolveShared>
__device__ __forceinline__ void solve_body(IcpDeviceState* __restrict__ st_, const SolveK* __restrict__ kp_,
                                           const double* __restrict__ partA, uint32_t nA, uint32_t strideA,
                                           const double* __restrict__ partB, uint32_t nB, uint32_t strideB,
                                           SH& sh, bool totA_ready = false, bool totB_ready = false) {
  double (*red)[64] = solve_red(sh);
  double* totA = sh.totA;
  double* totB = sh.totB;
  double (*sh_log)[6] = sh.sh_log;
  // the parameter block through the scalar path (uniform address, read-only), field by field: the 36-double prior is only
  // touched when present; the state block through a global-space pointer (mh_nn_device.h, G())
  const SolveK __attribute__((address_space(4)))& k = *(const SolveK __attribute__((address_space(4)))*)uniform_const_ptr(kp_);
  typedef typename std::conditional<LDS_STATE, IcpDeviceState __attribute__((address_space(3)))*, IcpDeviceState MH_AS_GLOBAL*>::type state_ptr;
  state_ptr const st = (state_ptr)st_;
  const int lane = threadIdx.x;
  if (nA)
    reduce_rows(partA, nA, strideA, kAccN, totA, red);
  if (nB)
    reduce_rows(partB, nB, strideB, kGenN, totB, red);
  double a[kAccN], gen[kGenN];
#pragma unroll
  for (int i = 0; i < kAccN; i++) a[i] = (nA || totA_ready) ? totA[i] : 0.0;
#pragma unroll
  for (int i = 0; i < kGenN; i++) gen[i] = (nB || totB_ready) ? totB[i] : 0.0;
  Pose Tc;
#pragma unroll
  for (int i = 0; i < 12; i++) Tc.m[i] = st->T[i];
  if (k.has_prior) {
    // e_p = log(T_prior^-1 (+) T); d e_p / d eps for T*exp(eps) by central differences, one lane
    // per perturbation (the exact derivative up to O(h^2); SURVEY App.B U9)
    Pose Pinv;
#pragma unroll
    for (int i = 0; i < 12; i++) Pinv.m[i] = k.prior_mean_inv[i];
    const Pose D = compose(Pinv, Tc);
    if (lane < 13) {
      double xi[6] = {0, 0, 0, 0, 0, 0};
      double h = 1e-6;
      asm volatile("" : "+v"(h));  // (opaque: inside k_icp16's loop the compiler otherwise computes the thirteen exp(xi) ahead of the loop and keeps them -- in scratch)
#pragma unroll
      for (int j = 0; j < 6; j++)
        if (lane < 12 && (lane >> 1) == j) xi[j] = (lane & 1) ? -h : h;
      const Pose Dp = compose(D, se3_exp(xi));
      double lg[6];
      se3_log(Dp, lg);
#pragma unroll
      for (int i = 0; i < 6; i++) sh_log[lane][i] = lg[i];
    }
    if (WAVE0) wave_sync_lds();
    else __syncthreads();
  }
  if (lane != 0) return;
  MH_PHASE(4);

  const uint32_t inner = st->inner;
  const uint32_t it = st->iter;
  double thr_next = 0.0, kparam_next = 0.0;  // fetched now, needed at the very end: two dependent loads off the tail
  if (it + 1 < k.max_iterations) {
    thr_next = k.thr[it + 1];
    kparam_next = k.kparam[it + 1];
  }
  const uint32_t n_pairs = (uint32_t)(a[17] + gen[28] + 0.5);
  if (inner == 0) {
    st->n_pairs = n_pairs;
    st->n_pairs_pl = (uint32_t)(gen[28] + 0.5);
    if (n_pairs == 0) {  // ICP::align: "if (pairings.empty()) NoPairings; break"
      st->term_reason = MH_TERM_NO_PAIRINGS;
      st->n_iterations = it;
      st->done = 1;
      return;
    }
  }
  // assemble the normal equations
  double H[36], g[6];
  for (int i = 0; i < 36; i++) H[i] = 0.0;
  H[0] = H[7] = H[14] = a[0];
  H[0 * 6 + 4] = a[3];  H[0 * 6 + 5] = -a[2];
  H[1 * 6 + 3] = -a[3]; H[1 * 6 + 5] = a[1];
  H[2 * 6 + 3] = a[2];  H[2 * 6 + 4] = -a[1];
  H[3 * 6 + 3] = a[4]; H[4 * 6 + 4] = a[5]; H[5 * 6 + 5] = a[6];
  H[3 * 6 + 4] = a[7]; H[3 * 6 + 5] = a[8]; H[4 * 6 + 5] = a[9];
  {
    int q = 0;
    for (int r = 0; r < 6; r++)
      for (int c = r; c < 6; c++) H[r * 6 + c] += gen[q++];
  }
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < r; c++) H[r * 6 + c] = H[c * 6 + r];
  for (int i = 0; i < 6; i++) g[i] = a[10 + i] + gen[21 + i];
  const double cost = a[16] + gen[27];
  if (k.has_prior) {
    double Jp[36];
    for (int j = 0; j < 6; j++)
      for (int i = 0; i < 6; i++) Jp[i * 6 + j] = (sh_log[2 * j][i] - sh_log[2 * j + 1][i]) / 2e-6;
    double JtL[36];
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) {
        double s = 0.0;
        for (int q = 0; q < 6; q++) s += Jp[q * 6 + i] * k.prior_info[q * 6 + j];
        JtL[i * 6 + j] = s;
      }
    for (int i = 0; i < 6; i++) {
      double s = 0.0;
      for (int q = 0; q < 6; q++) s += JtL[i * 6 + q] * sh_log[12][q];
      g[i] += s;
      for (int j = 0; j < 6; j++) {
        double h2 = 0.0;
        for (int q = 0; q < 6; q++) h2 += JtL[i * 6 + q] * Jp[q * 6 + j];
        H[i * 6 + j] += h2;
      }
    }
  }
  mh_gn_step* gt = (k.gn_trace && inner < kMaxGnTrace) ? &k.gn_trace[inner] : nullptr;
  if (gt) {
    for (int i = 0; i < 36; i++) gt->H[i] = H[i];
    for (int i = 0; i < 6; i++) { gt->g[i] = g[i]; gt->delta[i] = 0.0; }
    gt->err_norm_sqr = cost;
    for (int i = 0; i < 12; i++) gt->T_after[i] = Tc.m[i];
  }
  bool inner_done = false;
  MH_PHASE(5);
  if (sqrt(cost) <= k.max_cost) {
    inner_done = true;  // "target error" early exit, no solve (App.B U8)
  } else {
    double x[6], delta[6];
    if (!ldlt_solve6_spd(H, g, x) && !ldlt_solve6(H, g, x)) {  // (pivoted only for what the SPD fast path declines)
      st->solver_ok = 0;
      st->term_reason = MH_TERM_SOLVER_ERROR;
      st->n_iterations = it;
      st->done = 1;
      return;
    }
    double dn = 0.0;
    for (int i = 0; i < 6; i++) { delta[i] = -x[i]; dn += x[i] * x[i]; }
    MH_PHASE(6);
    Tc = compose(Tc, se3_exp(delta));  // T <- T (+) exp(delta)
    for (int i = 0; i < 12; i++) st->T[i] = Tc.m[i];
    MH_PHASE(7);
    st->n_solves += 1;
    if (gt) {
      for (int i = 0; i < 6; i++) gt->delta[i] = delta[i];
      for (int i = 0; i < 12; i++) gt->T_after[i] = Tc.m[i];
    }
    if (sqrt(dn) < k.min_delta) inner_done = true;
  }
  if (inner + 1 >= k.max_inner) inner_done = true;
  if (!inner_done) {
    st->inner = inner + 1;
    return;
  }
  // ---- end of ICP iteration `it` (tail of the loop body of ICP::align) ----
  st->inner = 0;
  MH_PHASE(8);
  Pose Tp;
  for (int i = 0; i < 12; i++) Tp.m[i] = st->T_prev[i];
  const Pose Drel = compose(inverse(Tp), Tc);
  // The stall test needs |log(Drel)|'s two halves -- a microsecond of the serial lane (atan2, tan, two square roots) -- only
  // where it can decide: |V^-1 t| >= |t| (V^-1 stretches what is perpendicular to the axis, keeps what is along it) and
  // theta^2 >= 2 (1 - cos theta), so a relative translation or a trace beyond the thresholds (with a margin far above the
  // rounding of either side) certifies "not stalled" without the logarithm.  Same decisions, same results.
  bool need_log = k.trace != nullptr;
  if (!need_log && !k.disable_stall) {
    const double tt = Drel.t(0) * Drel.t(0) + Drel.t(1) * Drel.t(1) + Drel.t(2) * Drel.t(2);
    const double one_minus_cos = 0.5 * (3.0 - (Drel.R(0, 0) + Drel.R(1, 1) + Drel.R(2, 2)));
    const bool moved = tt > k.min_step_trans * k.min_step_trans * (1.0 + 1e-6) ||
                       2.0 * one_minus_cos > k.min_step_rot * k.min_step_rot * (1.0 + 1e-6) + 1e-14;
    need_log = !moved;
  }
  if (need_log) {
    double d[6];
    se3_log(Drel, d);
    MH_PHASE(9);
    const double dtr = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const double drot = sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    if (k.trace) {
      mh_icp_iter* tr = &k.trace[it];
      for (int i = 0; i < 12; i++) tr->T[i] = Tc.m[i];
      tr->n_pairs = st->n_pairs;
      tr->threshold = k.thr ? k.thr[it] : 0.0;
      tr->kernel_param = k.kparam ? k.kparam[it] : 0.0;
      tr->delta_trans = dtr;
      tr->delta_rot = drot;
    }
    if (!k.disable_stall && dtr < k.min_step_trans && drot < k.min_step_rot) {
      st->term_reason = MH_TERM_STALLED;
      st->n_iterations = it;
      st->done = 1;
      return;
    }
  }
  if (k.hook_enabled) {
    // LidarOdometry.cpp:932-949: delta = currentSolution (-) checkpoint
    Pose Ci;
    for (int i = 0; i < 12; i++) Ci.m[i] = k.hook_chk_inv[i];
    const Pose S = compose(Ci, Tc);
    double w[3];
    so3_log(S, w);
    const double ht = sqrt(S.t(0) * S.t(0) + S.t(1) * S.t(1) + S.t(2) * S.t(2));
    const double hr = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (ht > k.hook_trans || hr > k.hook_rot) {
      st->term_reason = MH_TERM_HOOK_REQUEST;
      st->n_iterations = it;
      st->done = 1;
      return;
    }
  }
  for (int i = 0; i < 12; i++) st->T_prev[i] = Tc.m[i];
  st->iter = it + 1;
  if (it + 1 < k.max_iterations) {
    st->cur_thr2 = (float)(thr_next * thr_next);
    st->cur_kparam = kparam_next;
  }
  if (it + 1 >= k.max_iterations) {
    st->term_reason = MH_TERM_MAX_ITERATIONS;
    st->n_iterations = it + 1;
    st->done = 1;
  }
  MH_PHASE(10);
}

// first = 0: a solve of an inner Gauss-Newton step >= 1.  When the previous solve closed the ICP iteration early (step below
// min_delta, or the cost below max_cost: Solver_GaussNewton leaves its loop), the k_accum in front of this launch has
// skipped as well (same test) and the partials are stale: nothing to do.  (Found by tools/fuzz_batch.py: a converged
// alignment with the stall test off kept stepping on stale sums -- harmlessly small steps with k_accum's layout, garbage
// with the fused matchers' wider one.)
// One lane tells the host where the loop stands (system-scope store into page-locked host memory): the host then keeps
// only a couple of iterations queued ahead of the device instead of a predicted chunk with an idle tail (run_streaming).
__device__ __forceinline__ void publish_progress(const IcpDeviceState* __restrict__ st, const SolveK* __restrict__ kp) {
  if (threadIdx.x != 0) return;
  uint32_t* hp = kp->host_progress;
  if (!hp) return;
  const uint32_t v = (st->iter & 0x7FFFFFFFu) | (st->done ? 0x80000000u : 0u);
  __hip_atomic_store(hp, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ void k_solve_body(IcpDeviceState* __restrict__ st, const SolveK* __restrict__ kp,
                                                         const double* __restrict__ partA, uint32_t nA, uint32_t strideA,
                                                         const double* __restrict__ partB, uint32_t nB,
                                                         uint32_t strideB, uint32_t first) {
  __shared__ SolveShared sh;
  if (st->done) return;
  if (!first && st->inner == 0) return;  // (uniform: every lane reads the same word)
  solve_body(st, kp, partA, nA, strideA, partB, nB, strideB, sh);
  publish_progress(st, kp);
}

// ================================================================================================
// k_step16: the small-layer iteration with the solve CARRIED INTO THE NEXT LAUNCH (round 4).  The chain it replaces spent a
// launch of ONE workgroup on every Gauss-Newton step (k_match16 | k_accum_solve1 | k_accum_solve1, round 3: 28 us per ICP iteration of
// the real pipeline's 1.2-1.6 k-point layer).  Here every launch is the same kernel over the whole layer, and what it does
// is decided by the state block alone:
//   1. every workgroup copies the state block's head into LDS and, if a step is pending, closes it: the ordered sum of the
//      partials of the previous launch + solve_body -- all workgroups compute the same bits, nobody waits for a hand-over;
//      workgroup 0 writes the new state to ANOTHER state block (a ping-pong pair beside the canonical block: the one a launch
//      reads is never written by it) and publishes the progress word;
//   2. body: at the start of an ICP iteration (inner == 0) the row search of k_match16 for groups of 32 points, the pairings
//      stored and their Gauss-Newton sums written as one partial column per GROUP (also ping-pong: other workgroups may
//      still be reading the previous launch's); at an inner step the sums of the stored pairings under the new pose.
// An ICP iteration is max_inner launches (2 in the shipped pipelines) instead of 1 + max_inner, and a launch never idles
// because an iteration closed early: the next one simply starts in its place.  A workgroup takes the groups wg, wg + nw, ...:
// one group each for a single alignment; in a lock-step batch the host caps nw so that all jobs' workgroups are resident
// at once (512 threads x ~250 registers: one workgroup per CU) -- the columns, hence the sums and the result bit for bit, do
// not depend on nw.  `close_only` (one workgroup, in place into the canonical block): the pending step at the end of a
// chunk of launches.
// Phase stamps (tools/phase_probe.py, 1.4 k points, us): state into LDS 0.2, partials summed 1.4, assemble 0.4, LDLT 0.7,
// exp + compose 0.55, log 1.0, tail 0.6, state written 0.4, search + accumulate 1.1, sums 1.0 -- 7.5 of the ~13 us from one
// launch to the next; the rest is the launch.
// (Also built and measured in round 4, and removed: k_loop16, the whole loop in ONE launch -- the same body and solve per
// workgroup, a grid barrier between them (arrival counter + agent-scope loads of the partials; no cache invalidation, the map
// stays in the L2s).  Bit-identical, and slower: 15.8 us per Gauss-Newton step against 16.2 launch by launch in a 40-iteration
// fit, 0.594-0.611 against 0.544-0.551 ms of ICP per scan on the city drive (0.617 with an L2 write-back as the release).
// Crossing the XCDs costs what a launch boundary costs, and inside a loop the compiler hoists ~470 bytes per lane of lane
// masks, offset tables and literal constants into scratch.)
// (And: ONE launch per iteration, k_iter16 -- no sums cross workgroups at all: every workgroup accumulates ALL points of the layer
// for both Gauss-Newton steps of the iteration the previous launch matched, solves them on its own copy of the state, then
// searches its own groups.  Same trajectory file as k_step16's; 0.727-0.74 against 0.553 ms of ICP per scan: three rounds of
// agent-scope loads of the stored pairings per step and a called (not inlined: spills) solve cost more than the launch they
// save.  Removed.)
// (And: the covariance + the result written to the host's page-locked mirror by the launch that finds the loop finished --
// no covariance launches, no read-back copy, no event -- bit-identical to the three covariance kernels, 1536-1548 -> 1527-1573
// scans/s: the one workgroup that sums the whole layer takes what the launches took.  Removed.)
// ================================================================================================
constexpr uint32_t kStepPoints = kSolveThreads / 16;  // scan points (DPP rows) per group
constexpr uint32_t kStepRowsA = kAccN + 1;        // rows of one half of the point-to-point partials: the sums + the column's tag
constexpr uint32_t kStepRowsB = kGenN + 1;        // ... of the point-to-plane partials
constexpr uint32_t kStepMaxPoints = 8192;      // (above: k_match16<fused> | k_solve | k_accum | k_solve, then the quad matcher's chain)
constexpr uint32_t kStepMaxWorkgroups = 256;   // one per CU (512 threads x ~250 registers); beyond, workgroups take several groups
constexpr uint32_t kStateHeadDwords = (uint32_t)(offsetof(IcpDeviceState, cov) / 4);  // everything the loop touches
constexpr uint32_t kStateSerialDword = (uint32_t)(offsetof(IcpDeviceState, serial) / 4);
static_assert(kStateHeadDwords <= 64 && offsetof(IcpDeviceState, cov) % 8 == 0, "state head is copied by one wave");

// reduce_rows with agent-scope loads, all of a lane's loads issued before any sum, the sums in reduce_rows' order exactly (lane (row, g) adds columns g, g + G, ...: eight partial sums over the full rounds, the rest into
// the first, then the tree).  NVALS rows, up to kStepMaxPoints / kStepPoints columns.
template <int NVALS, int MAXCOLS = (int)(kStepMaxPoints / kStepPoints)>
struct RowLoads {
  static constexpr int kG = ((int)kSolveThreads / NVALS) > 64 ? 64 : ((int)kSolveThreads / NVALS);
  static constexpr int kL = (MAXCOLS + kG - 1) / kG;
  double v[kL];
};
template <int NVALS>
__device__ __forceinline__ void rows_issue(RowLoads<NVALS>& r, const double* part, uint32_t n, uint32_t stride) {
  constexpr int G = RowLoads<NVALS>::kG;
  const int row = (int)threadIdx.x / G, g = (int)threadIdx.x % G;
  const double* src = part + (size_t)(row < NVALS ? row : 0) * stride;
#pragma unroll
  for (int j = 0; j < RowLoads<NVALS>::kL; j++) {
    const uint32_t b = (uint32_t)(g + j * G);
    r.v[j] = (row < NVALS && b < n) ? __hip_atomic_load(src + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
  }
}
template <int NVALS, int MAXCOLS>
__device__ __forceinline__ void rows_finish(const RowLoads<NVALS, MAXCOLS>& r, uint32_t n, double* __restrict__ out, double (*red)[64]) {
  constexpr int G = RowLoads<NVALS, MAXCOLS>::kG;
  const int t = threadIdx.x, row = t / G, g = t % G;
  if (row < NVALS) {
    const uint32_t full = (n > (uint32_t)(g + 7 * G)) ? 1u + (n - (uint32_t)(g + 7 * G) - 1u) / (8u * G) : 0u;  // reduce_rows' rounds of eight
    double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < RowLoads<NVALS, MAXCOLS>::kL; j++) {
      if ((uint32_t)j < 8u * full) s[j % 8] += r.v[j];
      else if ((uint32_t)(g + j * G) < n) s[0] += r.v[j];
    }
    red[row][g] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  }
  __syncthreads();
  if (t < NVALS) {  // (all G reads first, then the additions in order: a read per addition costs its LDS latency G times over)
    double part[G];
#pragma unroll
    for (int q = 0; q < G; q++) part[q] = red[t][q];
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < G; q++) acc += part[q];
    out[t] = acc;
  }
  __syncthreads();
}

// Stored pairings cross launches of the k_step16 chain the way its state and partial sums do: agent-scope
// (write-through) stores, acknowledged before the group's partial column is tagged, and agent-scope loads by the launch that
// has seen the tag -- never answered from a stale L1 / L2 line, whichever XCD the reader runs on.
// (16 bytes in one sc1 access through a buffer descriptor -- an agent-scope __hip_atomic lowers to sc1 only up to 8 bytes, and
// 8-byte write-through stores cost 2.7x the 16-byte ones per byte -- with the compiler tracking the load like any other.)
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
struct AgentBuf {
  __amdgpu_buffer_rsrc_t rsrc;
};
__device__ __forceinline__ AgentBuf agent_buf(const void* base, uint32_t n_records) {
  const unsigned long long a = (unsigned long long)base;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  void* p = (void*)(((unsigned long long)hi << 32) | lo);
  AgentBuf b;
  b.rsrc = __builtin_amdgcn_make_buffer_rsrc(p, (short)0, (int)__builtin_amdgcn_readfirstlane((int)(n_records * 16u)), 0x00020000);
  return b;
}
__device__ __forceinline__ void store_agent_b128(const AgentBuf& b, uint32_t i, f32x4 v) {
  const u32x4v w = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(w, b.rsrc, (int)(i * 16u), 0, /*aux: sc1*/ 16);
}
__device__ __forceinline__ f32x4 load_agent_b128(const AgentBuf& b, uint32_t i) {
  const u32x4v w = __builtin_amdgcn_raw_buffer_load_b128(b.rsrc, (int)(i * 16u), 0, /*aux: sc1*/ 16);
  return (f32x4){__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w)};
}

template <bool PL>
__device__ __forceinline__ void k_step16_body(const IcpDeviceState* __restrict__ s_in, IcpDeviceState* s_out,
                                              IcpDeviceState* s_canon, const MatchK* __restrict__ kp,
                                              const SolveK* __restrict__ sk, const float* __restrict__ lx,
                                              const float* __restrict__ ly, const float* __restrict__ lz, uint32_t n,
                                              MapView map, float4* pair_q, uint32_t* pair_gidx, float4* pl_c, float4* pl_n,
                                              const double* __restrict__ partA_in, double* __restrict__ partA_out,
                                              const double* __restrict__ partB_in, double* __restrict__ partB_out,
                                              uint32_t ngroups, uint32_t nw, uint32_t close_only, uint32_t expect, uint32_t expect_rel) {
  __shared__ SolveShared sh;
  __shared__ __attribute__((aligned(8))) uint32_t lst_raw[kStateHeadDwords];
  __shared__ double rowsA[kAccN][kStepPoints + 1];
  __shared__ double rowsB[PL ? kGenN : 1][kStepPoints + 1];
  const AgentBuf b_pair = agent_buf(pair_q, n), b_plc = agent_buf(PL ? (const void*)pl_c : (const void*)pair_q, n),
                 b_pln = agent_buf(PL ? (const void*)pl_n : (const void*)pair_q, n);
  __shared__ uint32_t pair_acks;  // waves whose stores of the current and earlier groups are acknowledged
  uint32_t acks_wanted = 0;
  const uint32_t tid = threadIdx.x, wg = blockIdx.x;
  if (wg >= nw) return;
  if (tid == 0) pair_acks = 0;  // (barriers below before anybody counts)  // (lock-step batches: the grid is the largest job's)
  IcpDeviceState* const lst = reinterpret_cast<IcpDeviceState*>(lst_raw);
  // the point of this row is on its way before the state is looked at.  (Not so what the PREVIOUS launch stored for it: see below.)
  const uint32_t row = tid >> 4, r16 = tid & 15u;
  uint32_t g = wg;
  uint32_t i = g * kStepPoints + row;
  uint32_t ic = i < n ? i : n - 1;
  float x = G(lx)[ic], y = G(ly)[ic], z = G(lz)[ic];
  // (and the tag of "its" column of the partials: checked against the state's serial number below)
  static_assert(kStepMaxPoints / kStepPoints <= kSolveThreads, "a column per lane");
  const uint32_t col = tid < ngroups ? tid : 0u;
  double tag_a = __hip_atomic_load(partA_in + (size_t)kAccN * ngroups + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  double tag_b = PL ? __hip_atomic_load(partB_in + (size_t)kGenN * ngroups + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
  MH_PHASE(0);
  // The state block this launch is meant to read carries the serial number `expect` -- (the alignment's epoch << 22) + the
  // launches before this one; a launch replayed from a captured graph is told its place in the chunk and adds the serial number
  // the host wrote into the parameter block before the replay -- written by the upload or by workgroup 0 of the previous launch.  Anything else in the block is older (the previous alignment's, the launch before
  // last's: the same buffer): wait for the right one rather than act on it.
  if (expect_rel) expect += __hip_atomic_load(&sk->step_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (uint32_t spins = 0;; spins++) {
    if (tid < kStateHeadDwords) lst_raw[tid] = __hip_atomic_load(reinterpret_cast<const uint32_t*>(s_in) + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (lst->serial == expect) break;
    if (spins == (1u << 14)) {  // ~ tens of milliseconds: give up loudly (the host fails the alignment)
      if (tid == 0) {
        atomicAdd(&s_canon->handover_timeouts, 1u);
        if (atomicCAS(&s_canon->dbg[0], 0u, 1u) == 0u) {
          s_canon->dbg[1] = wg; s_canon->dbg[2] = tid; s_canon->dbg[3] = expect; s_canon->dbg[4] = lst->serial; s_canon->dbg[5] = ngroups;
        }
      }
      break;
    }
    __builtin_amdgcn_s_sleep(8);
    __syncthreads();  // (lst_raw is rewritten)
  }
  // (taken from the block NOW: workgroup 0 rewrites both words further down, behind a barrier -- a wave that looked at
  //  `pending` after thread 0 had set it for the NEXT launch waited for partial sums nobody had written: launch 0 of an
  //  alignment "gave up" in workgroup 0, a few times per thousand alignments, more under load)
  const uint32_t serial = lst->serial;  // what the columns this launch sums must be tagged with
  const uint32_t pending = lst->pending;
  if (lst->done) {  // the loop has ended (the canonical block has it): keep the ping-pong consistent, nothing else
    if (wg == 0 && tid < kStateHeadDwords)
      __hip_atomic_store(reinterpret_cast<uint32_t*>(s_out) + tid, tid == kStateSerialDword ? serial + 1u : lst_raw[tid], __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  MH_PHASE(1);
  f32x4 stored = (f32x4){0.f, 0.f, 0.f, 0.f}, stored_c = stored, stored_n = stored;
  uint32_t stored_g = kNoMatch;
  bool have_stored = false;
  if (pending) {
    // every column carries the serial number of the launch that wrote it, stored AFTER its sums were acknowledged: a column
    // that does not carry this launch's number yet has not arrived (never seen since the exchange is at agent scope; a lane
    // waits for its columns rather than sum what is not there)
    const double want = (double)serial;
    for (uint32_t spins = 0; tag_a != want || (PL && tag_b != want); spins++) {
      if (spins == (1u << 16)) {
        atomicAdd(&s_canon->handover_timeouts, 1u);
        if (atomicCAS(&s_canon->dbg[0], 0u, 2u) == 0u) {
          s_canon->dbg[1] = wg; s_canon->dbg[2] = tid; s_canon->dbg[3] = serial; s_canon->dbg[4] = (uint32_t)tag_a; s_canon->dbg[5] = ngroups;
          s_canon->dbg[6] = expect;
          s_canon->dbg[7] = (pending & 0xFFu) | ((lst->iter & 0xFFu) << 8) | ((lst->inner & 0xFFu) << 16) | ((lst->done & 0xFFu) << 24);
        }
        break;
      }
      __builtin_amdgcn_s_sleep(1);
      tag_a = __hip_atomic_load(partA_in + (size_t)kAccN * ngroups + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (PL) tag_b = __hip_atomic_load(partB_in + (size_t)kGenN * ngroups + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    // (the sums' loads were also tried ahead of the state block, with the tags: 0.594 against 0.563 ms of ICP per scan -- ten more
    //  loads in front of the one the launch waits for)
    RowLoads<kAccN> ra;
    RowLoads<PL ? kGenN : 1> rb;
    rows_issue<kAccN>(ra, partA_in, ngroups, ngroups);
    if (PL) rows_issue<PL ? kGenN : 1>(rb, partB_in, ngroups, ngroups);
    // every column of the previous launch is tagged: what its workgroups stored for their groups is acknowledged (the tag is
    // written after that) -- the stored pairings of this lane's point are requested now, at agent scope
    // and BEHIND the loads of the sums this launch waits for, and arrive while the sums are formed and the step is solved
    stored = load_agent_b128(b_pair, ic);
    stored_g = __hip_atomic_load(pair_gidx + ic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (PL) {
      stored_c = load_agent_b128(b_plc, ic);
      stored_n = load_agent_b128(b_pln, ic);
    }
    have_stored = true;
    rows_finish(ra, ngroups, sh.totA, sh.red);
    if (PL) rows_finish(rb, ngroups, sh.totB, sh.red);
    solve_body<true, false>(lst, sk, nullptr, 0u, 0u, nullptr, 0u, 0u, sh, true, PL);
    __syncthreads();
  }
  const uint32_t done = lst->done;
  const bool body = !done && !close_only;
  if (wg == 0) {
    __syncthreads();  // every wave has taken `serial` and `pending` from the block
    if (tid == 0) {
      lst->pending = body ? 1u : 0u;
      lst->serial = serial + 1u;
    }
    __syncthreads();
    if (tid < kStateHeadDwords) {
      const uint32_t w = lst_raw[tid];
      __hip_atomic_store(reinterpret_cast<uint32_t*>(s_out) + tid, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (done && s_canon != s_out) G(reinterpret_cast<uint32_t*>(s_canon))[tid] = w;
    }
    if (tid == 0) {
      uint32_t* hp = sk->host_progress;
      if (hp) __hip_atomic_store(hp, (lst->iter & 0x7FFFFFFFu) | (done ? 0x80000000u : 0u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  MH_PHASE(11);
  if (!body) return;
  MH_PHASE(14);
  // What the previous launch produced is consumed with care.  Under load from other streams (sixteen sequences in one
  // process) a launch can begin up to ~0.25 us before the end time stamp of its predecessor on the same stream (rocprofv3
  // kernel trace: 10 of 3464 k_step16_b dispatches), and data the predecessor wrote last was seen stale by loads issued
  // first thing: a partial column two launches old -- ulp-sized differences between a batch and the same alignment alone,
  // a few per 200-scan run, gone with ANY extra microsecond before the reads.  Hence: state and partials cross launches
  // through agent-scope stores and loads (write-through; never answered from a stale L2 line), every partial column is
  // tagged with its launch's serial number once its sums are acknowledged and a reader waits for the tag it expects (the
  // tags are on their way before the state is known: no extra round trip), and the stored pairings -- the previous pairing bounds the search at an iteration start and IS the pairing at an
  // inner step -- are read here, microseconds into the launch, not prefetched at its top.
  if (!have_stored) {  // (no step was pending: what is stored is at least two launches old)
    stored = load_agent_b128(b_pair, ic);
    stored_g = __hip_atomic_load(pair_gidx + ic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (PL) {
      stored_c = load_agent_b128(b_plc, ic);
      stored_n = load_agent_b128(b_pln, ic);
    }
  }

  typedef const MatchK __attribute__((address_space(4))) * cmatchk_ptr;
  typedef const double __attribute__((address_space(4))) * cf64_ptr;
  const cmatchk_ptr ck = (cmatchk_ptr)uniform_const_ptr(kp);
  const uint32_t inner = lst->inner, iter = lst->iter;
  double T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = lst->T[k];
  const float thr2 = lst->cur_thr2, ang2 = lst->cur_ang2;
  const double kparam = lst->cur_kparam;
  const uint32_t kernel = ck->kernel;
  for (;;) {  // the groups of this workgroup (workgroup-uniform trip count)
    Acc a;
    acc_zero(a);
    double v[PL ? kGenN : 1];
#pragma unroll
    for (int j = 0; j < (PL ? kGenN : 1); j++) v[j] = 0.0;
    if (i < n) {  // row-uniform
      f32x4 q = stored, bc = (f32x4){0.f, 0.f, 0.f, 0.f}, bn = (f32x4){0.f, 0.f, 0.f, 0.f};
      bool ok, okp = false;
      if (inner == 0) {  // (workgroup-uniform) a new ICP iteration: the matchers
        float px, py, pz;
        transform_point(T, x, y, z, px, py, pz);
        float bound0 = __builtin_inff();
        if (iter > 0 && !map.no_prev_bound && stored.w < __builtin_inff()) {
          const float dx = stored.x - px, dy = stored.y - py, dz = stored.z - pz;
          bound0 = (dx * dx + dy * dy) + dz * dz;  // the candidate arithmetic of the scans
        }
        const NNResult r = nn_search_row16(map, r16, px, py, pz, bound0);
        const float n2 = (px * px + py * py) + pz * pz;
        ok = r.found && (r.d2 < thr2 + ang2 * n2);
        if (PL) {  // Matcher_Point2Plane first (k_match16_body)
          const float pl_thr = (float)((cf64_ptr)uniform_const_ptr(ck->pl_thr))[iter];
          okp = pl_row_search(map, r16, px, py, pz, pl_thr, bc, bn);
          if (r16 == 0) {
            store_agent_b128(b_plc, i, (f32x4){bc.x, bc.y, bc.z, okp ? 1.f : 0.f});
            store_agent_b128(b_pln, i, (f32x4){bn.x, bn.y, bn.z, 0.f});
          }
          if (okp && ck->skip_pl_paired) ok = false;  // (the nearest point still goes to pair_q: it bounds the next search)
        }
        if (r16 == 0) {
          store_agent_b128(b_pair, i, (f32x4){r.pt.x, r.pt.y, r.pt.z, r.d2});
          __hip_atomic_store(pair_gidx + i, ok ? __float_as_uint(r.pt.w) : kNoMatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        q = (f32x4){r.pt.x, r.pt.y, r.pt.z, r.d2};
      } else {  // an inner Gauss-Newton step: the stored pairings under the new pose
        ok = stored_g != kNoMatch;
        if (PL) {
          bc = stored_c;
          bn = stored_n;
          okp = bc.w != 0.f;
        }
      }
      if (r16 == 0) {
        acc_pt2pt_masked(a, T, ok, x, y, z, q.x, q.y, q.z, kernel, kparam, ck->w_pt2pt);
        if (PL && okp)
          acc_pt2pl_rows(v, T, x, y, z, make_float4(bc.x, bc.y, bc.z, 1.f), make_float4(bn.x, bn.y, bn.z, 0.f), kernel, kparam,
                         ck->w_pt2pl);
      }
    }
    MH_PHASE(12);
    // 32 row leaders -> one partial per sum and group, fixed order
    if (r16 == 0) {
#pragma unroll
      for (int j = 0; j < kAccN; j++) rowsA[j][row] = a.v[j];
      if (PL) {
#pragma unroll
        for (int j = 0; j < kGenN; j++) rowsB[PL ? j : 0][row] = v[PL ? j : 0];
      }
    }
    __syncthreads();
    if (tid < kAccN) {
      double sum = rowsA[tid][0];
#pragma unroll
      for (int r = 1; r < (int)kStepPoints; r++) sum += rowsA[tid][r];
      __hip_atomic_store(partA_out + tid * ngroups + g, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (PL && tid >= 64 && tid < 64 + kGenN) {  // (the second wave: the 29 point-to-plane sums)
      const uint32_t t = tid - 64;
      double sum = rowsB[PL ? t : 0][0];
#pragma unroll
      for (int r = 1; r < (int)kStepPoints; r++) sum += rowsB[PL ? t : 0][r];
      __hip_atomic_store(partB_out + t * ngroups + g, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // A column is tagged once its sums AND the pairings every wave of the workgroup stored for the group are acknowledged (a
    // reader that has seen the tag reads them): every wave counts itself in when its own stores are -- the pairings were stored
    // before the sums were formed, so this adds nothing to what the tagging wave waits for anyway -- and the tagging wave waits
    // for the count.
    acks_wanted += kSolveThreads / 64u;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if ((tid & 63u) == 0u) __hip_atomic_fetch_add(&pair_acks, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (tid < 64 || (PL && tid < 128)) {  // (the first wave holds the 18 sums: lane 18 tags the column; the second wave's lane 29 the other kind's)
      while (__hip_atomic_load(&pair_acks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < acks_wanted) __builtin_amdgcn_s_sleep(1);
      if (tid == kAccN) __hip_atomic_store(partA_out + (size_t)kAccN * ngroups + g, (double)(serial + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (PL && tid == 64 + kGenN) __hip_atomic_store(partB_out + (size_t)kGenN * ngroups + g, (double)(serial + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    g += nw;
    if (g >= ngroups) break;
    i = g * kStepPoints + row;
    ic = i < n ? i : n - 1;
    x = G(lx)[ic]; y = G(ly)[ic]; z = G(lz)[ic];
    stored = load_agent_b128(b_pair, ic);
    stored_g = __hip_atomic_load(pair_gidx + ic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (PL) {
      stored_c = load_agent_b128(b_plc, ic);
      stored_n = load_agent_b128(b_pln, ic);
    }
    __syncthreads();  // (the row buffers are reused)
  }
  MH_PHASE(13);
}
template <bool PL>
__global__ __launch_bounds__(kSolveThreads) void k_step16(const IcpDeviceState* __restrict__ s_in, IcpDeviceState* s_out,
                                                          IcpDeviceState* s_canon, const MatchK* __restrict__ kp,
                                                          const SolveK* __restrict__ sk, const float* __restrict__ lx,
                                                          const float* __restrict__ ly, const float* __restrict__ lz,
                                                          uint32_t n, MapView map, float4* pair_q, uint32_t* pair_gidx,
                                                          float4* pl_c, float4* pl_n, const double* __restrict__ partA_in,
                                                          double* __restrict__ partA_out, const double* __restrict__ partB_in,
                                                          double* __restrict__ partB_out, uint32_t ngroups, uint32_t close_only,
                                                          uint32_t expect, uint32_t expect_rel) {
  k_step16_body<PL>(s_in, s_out, s_canon, kp, sk, lx, ly, lz, n, map, pair_q, pair_gidx, pl_c, pl_n, partA_in, partA_out, partB_in,
                    partB_out, ngroups, gridDim.x, close_only, expect, expect_rel);
}
// in lock step: blockIdx.y = job; `par`: which state block / partials half this launch reads; gridDim.x: the host's cap on a
// job's workgroups
template <bool PL>
__global__ __launch_bounds__(kSolveThreads) void k_step16_b(const BatchJob* __restrict__ jobs, uint32_t src, uint32_t par, uint32_t close_only,
                                                            uint32_t launch_index) {
  const BatchJob& j = jobs[blockIdx.y];
  const uint32_t ng0 = (j.n + kStepPoints - 1) / kStepPoints;
  const uint32_t ngroups = ng0 ? ng0 : 1u;
  IcpDeviceState* const S[3] = {j.st_b, reinterpret_cast<IcpDeviceState*>(reinterpret_cast<char*>(j.st_b) + 256), j.st};
  const uint32_t dst = close_only ? 2u : (src == 2u ? 0u : (src ^ 1u));
  double* const pa[2] = {j.part, j.part + (size_t)kStepRowsA * ngroups};
  double* const pb[2] = {j.partb, j.partb ? j.partb + (size_t)kStepRowsB * ngroups : nullptr};
  k_step16_body<PL>(S[src], S[dst], S[2], j.mk, j.sk, j.lx, j.ly, j.lz, j.n, j.map, j.pair_q, j.pair_gidx,
                    j.pl_c, j.pl_n, pa[par], pa[par ^ 1u], pb[par], pb[par ^ 1u], ngroups, ngroups < gridDim.x ? ngroups : gridDim.x,
                    close_only, j.serial_base + launch_index, 0u);
}

// ================================================================================================
// k_icp16: the small layer's WHOLE loop in one launch (round 5) -- the k_step16 chain without its launch boundaries.
// One workgroup per group of 32 points, all of them resident for the duration (the host admits a loop only while the
// workgroups of all running loops fit the part's CUs, and falls back to the chain otherwise or when a workgroup gives up
// waiting); every workgroup keeps its own copy of the state block in LDS and closes every Gauss-Newton step itself -- the
// same ordered sums and the same solve_body as k_step16, the same bits -- so that only the partial sums cross workgroups:
//   body (search or re-accumulate, the pairings of the group stay in registers) -> the group's column of sums, every sum a
//   16-byte entry {value, serial number, check word} in ONE agent-scope store -> every workgroup loads all columns of the
//   step and retries the entries that do not carry the step's serial number yet (no separate tag: one round trip instead of
//   store | acknowledge | tag | poll | load) -> ordered sums -> solve -> next body.
// tools/xcd_exchange.hip prices the exchange alone: 2.1 us for 44 workgroups x 18 sums (3.2 with the 29 plane sums; 3.1 / 4+
// with a tag per column), against ~4.5 us of launch boundary + ~2 us of tagged exchange per k_step16 launch.  (The same tool:
// workgroups of a launch are dealt to the XCDs round-robin, blockIdx % 8, but sc0 loads do not bypass the L1 -- an exchange
// confined to one XCD's L2 has no cheaper load than the agent-scope one, and 32 CUs would hold 32 groups only.)
// The entries ping-pong between two halves by step parity: a workgroup writes step s + 2's entries after it has summed
// step s + 1, which every workgroup wrote after reading step s.  Serial numbers never repeat within a context (a counter
// advanced by every loop's step budget), so an entry of an earlier alignment is never taken for the current one.
// ================================================================================================
#ifndef MH_LOOP_MAX_GROUPS
#define MH_LOOP_MAX_GROUPS 64
#endif
constexpr uint32_t kLoopMaxGroups = MH_LOOP_MAX_GROUPS;   // workgroups of one k_icp16 loop: layers up to 32 x this many points (beyond: the chain)
constexpr uint32_t kLwMaxGroups = 128;                    // columns of one k_icpw loop (mh_loop_wave.h): layers up to 4096 points
constexpr uint32_t kLoopRowStride = kLwMaxGroups > kLoopMaxGroups ? kLwMaxGroups : kLoopMaxGroups;   // entries from one sum's row to the next
constexpr size_t kLoopExchangeBytes = 2 * (size_t)(kAccN + kGenN) * kLoopRowStride * 16;

__device__ __forceinline__ void cov_prepare_lane(const Pose& Tc, int j, double hx, double ha, double* out);  // (below)

template <int NVALS>
__device__ __forceinline__ void loop_rows_fetch(RowLoads<NVALS, (int)kLoopMaxGroups>& r, const AgentBuf& x, uint32_t base, uint32_t n,
                                                uint32_t serial, uint32_t* gave_up) {
  typedef RowLoads<NVALS, (int)kLoopMaxGroups> RL;
  const int row = (int)threadIdx.x / RL::kG, g = (int)threadIdx.x % RL::kG;
  uint32_t need = 0;
#pragma unroll
  for (int j = 0; j < RL::kL; j++) {
    r.v[j] = 0.0;
    if (row < NVALS && (uint32_t)(g + j * RL::kG) < n) need |= 1u << j;
  }
  const uint32_t e0 = base + (uint32_t)(row < NVALS ? row : 0) * kLoopRowStride + (uint32_t)g;
  for (uint32_t spins = 0; need; spins++) {
    u32x4v w[RL::kL];
#pragma unroll
    for (int j = 0; j < RL::kL; j++)
      if ((need >> j) & 1u) w[j] = __builtin_amdgcn_raw_buffer_load_b128(x.rsrc, (int)((e0 + (uint32_t)(j * RL::kG)) * 16u), 0, /*aux: sc1*/ 16);
#pragma unroll
    for (int j = 0; j < RL::kL; j++)
      if (((need >> j) & 1u) && w[j].z == serial && w[j].w == (w[j].x ^ w[j].y ^ w[j].z)) {
        r.v[j] = __hiloint2double((int)w[j].y, (int)w[j].x);
        need &= ~(1u << j);
      }
    if (!need) break;
    if (spins == (1u << 16) || __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) {  // ~0.1 s: give up loudly
      __hip_atomic_store(gave_up, 1u + (uint32_t)row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}
__device__ __forceinline__ void loop_entry_store(const AgentBuf& x, uint32_t e, double v, uint32_t serial) {
  const uint32_t lo = (uint32_t)__double2loint(v), hi = (uint32_t)__double2hiint(v);
  const u32x4v w = {lo, hi, serial, lo ^ hi ^ serial};
  __builtin_amdgcn_raw_buffer_store_b128(w, x.rsrc, (int)(e * 16u), 0, /*aux: sc1*/ 16);
}

template <bool PL>
__global__ __launch_bounds__(kSolveThreads) void k_icp16(IcpDeviceState* s_canon, const MatchK* __restrict__ kp,
                                                         const SolveK* __restrict__ sk, const float* __restrict__ lx,
                                                         const float* __restrict__ ly, const float* __restrict__ lz, uint32_t n,
                                                         MapView map, float4* pair_q, uint32_t* pair_gidx, float4* pl_c, float4* pl_n,
                                                         void* xa, void* xb, uint32_t ngroups, uint32_t serial0, uint32_t max_steps,
                                                         uint32_t want_cov) {
  __shared__ SolveShared sh;
  __shared__ __attribute__((aligned(8))) uint32_t lst_raw[kStateHeadDwords];
  __shared__ double rowsA[kAccN][kStepPoints + 1];
  __shared__ double rowsB[PL ? kGenN : 1][kStepPoints + 1];
  __shared__ uint32_t gave_up;
  const uint32_t tid = threadIdx.x, g = blockIdx.x;
  if (g >= ngroups) return;
  IcpDeviceState* const lst = reinterpret_cast<IcpDeviceState*>(lst_raw);
  const AgentBuf bxa = agent_buf(xa, 2u * kAccN * kLoopRowStride), bxb = agent_buf(PL ? xb : xa, 2u * (PL ? kGenN : kAccN) * kLoopRowStride);
  const uint32_t row = tid >> 4, r16 = tid & 15u;
  const uint32_t i = g * kStepPoints + row, ic = i < n ? i : n - 1;
  const float x = G(lx)[ic], y = G(ly)[ic], z = G(lz)[ic];
  if (tid < kStateHeadDwords) lst_raw[tid] = G(reinterpret_cast<const uint32_t*>(s_canon))[tid];  // (uploaded before the launch)
  if (tid == 0) gave_up = 0;
  __syncthreads();
  typedef const MatchK __attribute__((address_space(4))) * cmatchk_ptr;
  typedef const double __attribute__((address_space(4))) * cf64_ptr;
  const cmatchk_ptr ck = (cmatchk_ptr)uniform_const_ptr(kp);
  const uint32_t kernel = ck->kernel;
  // the pairing of this row's point: found at an iteration's start, used by its inner steps and as the next search's bound
  f32x4 q = (f32x4){0.f, 0.f, 0.f, __builtin_inff()}, bc = (f32x4){0.f, 0.f, 0.f, 0.f}, bn = (f32x4){0.f, 0.f, 0.f, 0.f};
  bool ok = false, okp = false;
  uint32_t step = 0;
  MH_LOOP_STAMPS;
#pragma nounroll
  for (;; step++) {
    MH_LOOP_STAMP(0);
    if (lst->pending) {  // the sums of step - 1 (serial0 + step), every workgroup for itself
      const uint32_t half = (step - 1u) & 1u;
      RowLoads<kAccN, (int)kLoopMaxGroups> ra;
      RowLoads<PL ? kGenN : 1, (int)kLoopMaxGroups> rb;
      loop_rows_fetch<kAccN>(ra, bxa, half * kAccN * kLoopRowStride, ngroups, serial0 + step, &gave_up);
      if (PL) loop_rows_fetch<PL ? kGenN : 1>(rb, bxb, half * kGenN * kLoopRowStride, ngroups, serial0 + step, &gave_up);
      MH_LOOP_STAMP(1);
      rows_finish(ra, ngroups, sh.totA, sh.red);
      if (PL) rows_finish(rb, ngroups, sh.totB, sh.red);
      MH_LOOP_STAMP(2);
      if (gave_up) break;  // (behind rows_finish' barriers: the same in every wave)
      solve_body<true, false>(lst, sk, nullptr, 0u, 0u, nullptr, 0u, 0u, sh, true, PL);
      if (tid == 0) lst->pending = 0u;
      __syncthreads();
      MH_LOOP_STAMP(3);
    }
    if (lst->done || step >= max_steps) break;
    const uint32_t inner = lst->inner, iter = lst->iter;
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = lst->T[k];
    const float thr2 = lst->cur_thr2, ang2 = lst->cur_ang2;
    const double kparam = lst->cur_kparam;
    Acc a;
    acc_zero(a);
    double v[PL ? kGenN : 1];
#pragma unroll
    for (int j = 0; j < (PL ? kGenN : 1); j++) v[j] = 0.0;
    if (i < n) {  // row-uniform
      if (inner == 0) {  // (workgroup-uniform) a new ICP iteration: the matchers, exactly k_step16's
        float px, py, pz;
        transform_point(T, x, y, z, px, py, pz);
        float bound0 = __builtin_inff();
        if (iter > 0 && !map.no_prev_bound && q.w < __builtin_inff()) {
          const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
          bound0 = (dx * dx + dy * dy) + dz * dz;  // the candidate arithmetic of the scans
        }
        MH_LOOP_STAMP(6);
        const NNResult r = nn_search_row16(map, r16, px, py, pz, bound0);
        MH_LOOP_STAMP(7);
        const float n2 = (px * px + py * py) + pz * pz;
        ok = r.found && (r.d2 < thr2 + ang2 * n2);
        if (PL) {  // Matcher_Point2Plane first (k_match16_body)
          const float pl_thr = (float)((cf64_ptr)uniform_const_ptr(ck->pl_thr))[iter];
          okp = pl_row_search(map, r16, px, py, pz, pl_thr, bc, bn);
          if (r16 == 0) {  // (read by the covariance kernels and the pairing export once the loop has ended)
            pl_c[i] = make_float4(bc.x, bc.y, bc.z, okp ? 1.f : 0.f);
            pl_n[i] = make_float4(bn.x, bn.y, bn.z, 0.f);
          }
          if (okp && ck->skip_pl_paired) ok = false;  // (the nearest point still bounds the next search)
        }
        if (r16 == 0) {
          pair_q[i] = make_float4(r.pt.x, r.pt.y, r.pt.z, r.d2);
          G(pair_gidx)[i] = ok ? __float_as_uint(r.pt.w) : kNoMatch;
        }
        q = (f32x4){r.pt.x, r.pt.y, r.pt.z, r.d2};
      }
      if (r16 == 0) {
        acc_pt2pt_masked(a, T, ok, x, y, z, q.x, q.y, q.z, kernel, kparam, ck->w_pt2pt);
        if (PL && okp)
          acc_pt2pl_rows(v, T, x, y, z, make_float4(bc.x, bc.y, bc.z, 1.f), make_float4(bn.x, bn.y, bn.z, 0.f), kernel, kparam,
                         ck->w_pt2pl);
      }
    }
    if (r16 == 0) {
#pragma unroll
      for (int j = 0; j < kAccN; j++) rowsA[j][row] = a.v[j];
      if (PL) {
#pragma unroll
        for (int j = 0; j < kGenN; j++) rowsB[PL ? j : 0][row] = v[PL ? j : 0];
      }
    }
    MH_LOOP_STAMP(8);
    __syncthreads();
    MH_LOOP_STAMP(4);
    const uint32_t out = step & 1u;
    if (tid < kAccN) {
      double sum = rowsA[tid][0];
#pragma unroll
      for (int r = 1; r < (int)kStepPoints; r++) sum += rowsA[tid][r];
      loop_entry_store(bxa, (out * kAccN + tid) * kLoopRowStride + g, sum, serial0 + step + 1u);
    }
    if (PL && tid >= 64 && tid < 64 + kGenN) {  // (the second wave: the 29 point-to-plane sums)
      const uint32_t t = tid - 64;
      double sum = rowsB[PL ? t : 0][0];
#pragma unroll
      for (int r = 1; r < (int)kStepPoints; r++) sum += rowsB[PL ? t : 0][r];
      loop_entry_store(bxb, (out * kGenN + t) * kLoopRowStride + g, sum, serial0 + step + 1u);
    }
    if (tid == 0) lst->pending = 1u;
    __syncthreads();
    MH_LOOP_STAMP(5);
  }
  MH_LOOP_STAMPS_OUT(step);
  if (gave_up) {  // the canonical block keeps done == 0: the host runs the alignment again, launch by launch
    if (tid == 0) {
      atomicAdd(&s_canon->handover_timeouts, 1u);
      if (atomicCAS(&s_canon->dbg[0], 0u, 3u) == 0u) {
        s_canon->dbg[1] = g; s_canon->dbg[2] = gave_up - 1u; s_canon->dbg[3] = serial0 + step; s_canon->dbg[4] = step; s_canon->dbg[5] = ngroups;
      }
    }
    return;
  }
  if (g == 0) {
    if (tid < kStateHeadDwords) G(reinterpret_cast<uint32_t*>(s_canon))[tid] = lst_raw[tid];
    if (want_cov && lst->done && tid < 6) {  // k_cov_prepare's six lanes: the covariance chain that follows starts at k_cov_accum
      Pose Tc;
#pragma unroll
      for (int k = 0; k < 12; k++) Tc.m[k] = lst->T[k];
      double out[12];
      cov_prepare_lane(Tc, (int)tid, sk->cov_hx, sk->cov_ha, out);
#pragma unroll
      for (int k = 0; k < 12; k++) s_canon->covD[tid * 12 + k] = out[k];
    }
  }
}

// k_icp16_b: the same loop for the jobs of a lock-step group, side by side in ONE launch (blockIdx.y = job) -- every job's
// workgroups exchange among themselves only; a workgroup takes the groups x, x + nw, ... of its job (the host caps nw so that the
// workgroups of ALL jobs are resident together), one column of sums per GROUP as everywhere: the same bits.  The pairings of a
// workgroup's groups wait in LDS (up to kLoopGroupsPerWg groups of 32 rows) instead of registers.
constexpr uint32_t kLoopGroupsPerWg = 8;
template <bool PL>
__device__ __forceinline__ void icp16_multi_body(const BatchJob& j) {
  __shared__ SolveShared sh;
  __shared__ __attribute__((aligned(8))) uint32_t lst_raw[kStateHeadDwords];
  __shared__ double rowsA[kAccN][kStepPoints + 1];
  __shared__ double rowsB[PL ? kGenN : 1][kStepPoints + 1];
  __shared__ f32x4 keep_q[kLoopGroupsPerWg][kStepPoints];
  __shared__ f32x4 keep_c[PL ? kLoopGroupsPerWg : 1][kStepPoints], keep_n[PL ? kLoopGroupsPerWg : 1][kStepPoints];
  __shared__ uint32_t keep_ok[kLoopGroupsPerWg][kStepPoints];  // bit 0: point pairing accepted, bit 1: plane pairing
  __shared__ uint32_t gave_up;
  const uint32_t n = j.n;
  const uint32_t ng0 = (n + kStepPoints - 1) / kStepPoints, ngroups = ng0 ? ng0 : 1u;
  const uint32_t nw = ngroups < gridDim.x ? ngroups : gridDim.x;
  const uint32_t tid = threadIdx.x, wg = blockIdx.x;
  if (wg >= nw || n == 0) return;
  IcpDeviceState* const s_canon = j.st;
  const SolveK* const sk = j.sk;
  const MapView map = j.map;
  IcpDeviceState* const lst = reinterpret_cast<IcpDeviceState*>(lst_raw);
  const AgentBuf bxa = agent_buf(j.loop_xa, 2u * kAccN * kLoopRowStride), bxb = agent_buf(PL ? j.loop_xb : j.loop_xa, 2u * (PL ? kGenN : kAccN) * kLoopRowStride);
  const uint32_t row = tid >> 4, r16 = tid & 15u;
  const uint32_t serial0 = j.loop_serial0;
  if (tid < kStateHeadDwords) lst_raw[tid] = G(reinterpret_cast<const uint32_t*>(s_canon))[tid];  // (scattered before the launch)
  if (tid == 0) gave_up = 0;
  __syncthreads();
  if (lst->done) return;  // (a job that was finished before the batch began: nothing to do)
  typedef const MatchK __attribute__((address_space(4))) * cmatchk_ptr;
  typedef const double __attribute__((address_space(4))) * cf64_ptr;
  const cmatchk_ptr ck = (cmatchk_ptr)uniform_const_ptr(j.mk);
  const uint32_t kernel = ck->kernel;
  const uint32_t max_steps = j.loop_pad ? 1u : sk->max_iterations * sk->max_inner + 1u;  // (loop_pad: MH_LOOP16_TEST_ABANDON, the loop is cut short)
  uint32_t step = 0;
#pragma nounroll
  for (;; step++) {
    if (lst->pending) {
      const uint32_t half = (step - 1u) & 1u;
      RowLoads<kAccN, (int)kLoopMaxGroups> ra;
      RowLoads<PL ? kGenN : 1, (int)kLoopMaxGroups> rb;
      loop_rows_fetch<kAccN>(ra, bxa, half * kAccN * kLoopRowStride, ngroups, serial0 + step, &gave_up);
      if (PL) loop_rows_fetch<PL ? kGenN : 1>(rb, bxb, half * kGenN * kLoopRowStride, ngroups, serial0 + step, &gave_up);
      rows_finish(ra, ngroups, sh.totA, sh.red);
      if (PL) rows_finish(rb, ngroups, sh.totB, sh.red);
      if (gave_up) break;
      solve_body<true, false>(lst, sk, nullptr, 0u, 0u, nullptr, 0u, 0u, sh, true, PL);
      if (tid == 0) lst->pending = 0u;
      __syncthreads();
    }
    if (lst->done || step >= max_steps) break;
    const uint32_t inner = lst->inner, iter = lst->iter;
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = lst->T[k];
    const float thr2 = lst->cur_thr2, ang2 = lst->cur_ang2;
    const double kparam = lst->cur_kparam;
    const uint32_t out = step & 1u;
    uint32_t slot = 0;
#pragma nounroll
    for (uint32_t g = wg; g < ngroups; g += nw, slot++) {  // (workgroup-uniform trip count)
      const uint32_t i = g * kStepPoints + row, ic = i < n ? i : n - 1;
      const float x = G(j.lx)[ic], y = G(j.ly)[ic], z = G(j.lz)[ic];
      Acc a;
      acc_zero(a);
      double v[PL ? kGenN : 1];
#pragma unroll
      for (int q = 0; q < (PL ? kGenN : 1); q++) v[q] = 0.0;
      if (i < n) {
        f32x4 q = keep_q[slot][row], bc = (f32x4){0.f, 0.f, 0.f, 0.f}, bn = (f32x4){0.f, 0.f, 0.f, 0.f};
        bool ok, okp = false;
        if (inner == 0) {
          float px, py, pz;
          transform_point(T, x, y, z, px, py, pz);
          float bound0 = __builtin_inff();
          if (iter > 0 && !map.no_prev_bound && q.w < __builtin_inff()) {
            const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
            bound0 = (dx * dx + dy * dy) + dz * dz;
          }
          const NNResult r = nn_search_row16(map, r16, px, py, pz, bound0);
          const float n2 = (px * px + py * py) + pz * pz;
          ok = r.found && (r.d2 < thr2 + ang2 * n2);
          if (PL) {
            const float pl_thr = (float)((cf64_ptr)uniform_const_ptr(ck->pl_thr))[iter];
            okp = pl_row_search(map, r16, px, py, pz, pl_thr, bc, bn);
            if (r16 == 0) {
              j.pl_c[i] = make_float4(bc.x, bc.y, bc.z, okp ? 1.f : 0.f);
              j.pl_n[i] = make_float4(bn.x, bn.y, bn.z, 0.f);
              keep_c[PL ? slot : 0][row] = bc;
              keep_n[PL ? slot : 0][row] = bn;
            }
            if (okp && ck->skip_pl_paired) ok = false;
          }
          q = (f32x4){r.pt.x, r.pt.y, r.pt.z, r.d2};
          if (r16 == 0) {
            j.pair_q[i] = make_float4(r.pt.x, r.pt.y, r.pt.z, r.d2);
            G(j.pair_gidx)[i] = ok ? __float_as_uint(r.pt.w) : kNoMatch;
            keep_q[slot][row] = q;
            keep_ok[slot][row] = (ok ? 1u : 0u) | (okp ? 2u : 0u);
          }
        } else {
          const uint32_t f = keep_ok[slot][row];
          ok = (f & 1u) != 0u;
          okp = (f & 2u) != 0u;
          if (PL) {
            bc = keep_c[PL ? slot : 0][row];
            bn = keep_n[PL ? slot : 0][row];
          }
        }
        if (r16 == 0) {
          acc_pt2pt_masked(a, T, ok, x, y, z, q.x, q.y, q.z, kernel, kparam, ck->w_pt2pt);
          if (PL && okp)
            acc_pt2pl_rows(v, T, x, y, z, make_float4(bc.x, bc.y, bc.z, 1.f), make_float4(bn.x, bn.y, bn.z, 0.f), kernel, kparam,
                           ck->w_pt2pl);
        }
      }
      if (r16 == 0) {
#pragma unroll
        for (int q = 0; q < kAccN; q++) rowsA[q][row] = a.v[q];
        if (PL) {
#pragma unroll
          for (int q = 0; q < kGenN; q++) rowsB[PL ? q : 0][row] = v[PL ? q : 0];
        }
      }
      __syncthreads();
      if (tid < kAccN) {
        double sum = rowsA[tid][0];
#pragma unroll
        for (int r = 1; r < (int)kStepPoints; r++) sum += rowsA[tid][r];
        loop_entry_store(bxa, (out * kAccN + tid) * kLoopRowStride + g, sum, serial0 + step + 1u);
      }
      if (PL && tid >= 64 && tid < 64 + kGenN) {
        const uint32_t t = tid - 64;
        double sum = rowsB[PL ? t : 0][0];
#pragma unroll
        for (int r = 1; r < (int)kStepPoints; r++) sum += rowsB[PL ? t : 0][r];
        loop_entry_store(bxb, (out * kGenN + t) * kLoopRowStride + g, sum, serial0 + step + 1u);
      }
      __syncthreads();  // (the row buffers are reused by the workgroup's next group)
    }
    if (tid == 0) lst->pending = 1u;
    __syncthreads();
  }
  if (gave_up) {
    if (tid == 0) {
      atomicAdd(&s_canon->handover_timeouts, 1u);
      if (atomicCAS(&s_canon->dbg[0], 0u, 4u) == 0u) {
        s_canon->dbg[1] = wg; s_canon->dbg[2] = gave_up - 1u; s_canon->dbg[3] = serial0 + step; s_canon->dbg[4] = step; s_canon->dbg[5] = ngroups;
      }
    }
    return;
  }
  if (wg == 0 && tid < kStateHeadDwords) G(reinterpret_cast<uint32_t*>(s_canon))[tid] = lst_raw[tid];
}
template <bool PL>
__global__ __launch_bounds__(kSolveThreads) void k_icp16_b(const BatchJob* __restrict__ jobs) {
  icp16_multi_body<PL>(jobs[blockIdx.y]);
}

#include "mh_loop_wave.h"  // k_icpw: the same loop with the plan / scan search, 128 points per workgroup of 256 lanes

// ================================================================================================
// Covariance (mp2p_icp::covariance [U]): A = d residuals / d (x,y,z,yaw,pitch,roll), cov = (A^T A)^-1
// ================================================================================================
// (Round 3 built two further fusions of the small-layer chain -- k_accum_solveN: all inner Gauss-Newton steps of an iteration
// in one launch; k_icp_persist: the WHOLE alignment of a layer of <= 2048 points in one workgroup and one launch -- bit-exact,
// parity-tested, and slower: 0.90 against 0.815 ms of ICP per scan, and 3.5 against 0.81 ms (one CU cannot supply the search's
// memory-level parallelism).  Both were selectable through MH_FUSED_INNER / MH_PERSIST until round 4 removed them; the
// measurements are in profiles/r03_persist_kernel.md and DESIGN.md section 3, the code in the history (round 3's last commit).)

constexpr int kCovN = 22;  // 21 upper-triangle + count

// column j of d T / d (x, y, z, yaw, pitch, roll) by central differences -> out[12]
__device__ __forceinline__ void cov_prepare_lane(const Pose& Tc, int j, double hx, double ha, double* out) {
  double v[6];
  pose_to_ypr(Tc, v);
  const double h = j < 3 ? hx : ha;
  double vp[6], vm[6];
  for (int i = 0; i < 6; i++) { vp[i] = v[i]; vm[i] = v[i]; }
  vp[j] += h;
  vm[j] -= h;
  const Pose P = pose_from_ypr(vp), M = pose_from_ypr(vm);
  for (int i = 0; i < 12; i++) out[i] = (P.m[i] - M.m[i]) / (2.0 * h);
}
// A^T A rows of one point-to-point pairing (3 residual rows) / of one point-to-plane pairing (1 row)
__device__ __forceinline__ void cov_rows_point(const double* sD, double x, double y, double z, double* v) {
  double A[3][6];
#pragma unroll
  for (int j = 0; j < 6; j++)
#pragma unroll
    for (int r = 0; r < 3; r++)
      A[r][j] = sD[j * 12 + r * 4] * x + sD[j * 12 + r * 4 + 1] * y + sD[j * 12 + r * 4 + 2] * z + sD[j * 12 + r * 4 + 3];
  int q = 0;
#pragma unroll
  for (int a = 0; a < 6; a++)
#pragma unroll
    for (int b = a; b < 6; b++) v[q++] = A[0][a] * A[0][b] + A[1][a] * A[1][b] + A[2][a] * A[2][b];
  v[21] = 1.0;
}
__device__ __forceinline__ void cov_rows_plane(const double* sD, double x, double y, double z, double nx, double ny, double nz,
                                               double* v) {
  double A[6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double r[3];
#pragma unroll
    for (int q = 0; q < 3; q++)
      r[q] = sD[j * 12 + q * 4] * x + sD[j * 12 + q * 4 + 1] * y + sD[j * 12 + q * 4 + 2] * z + sD[j * 12 + q * 4 + 3];
    A[j] = nx * r[0] + ny * r[1] + nz * r[2];
  }
  int q = 0;
#pragma unroll
  for (int a = 0; a < 6; a++)
#pragma unroll
    for (int b = a; b < 6; b++) v[q++] = A[a] * A[b];
  v[21] = 1.0;
}
// (A^T A)^-1 from the 21 + 1 sums; diag(1e6) when there is nothing to invert
__device__ __forceinline__ void cov_from_sums(const double* a, double* out36) {
  double AtA[36], cov[36];
  int q = 0;
  for (int r = 0; r < 6; r++)
    for (int c = r; c < 6; c++) {
      AtA[r * 6 + c] = a[q];
      AtA[c * 6 + r] = a[q];
      q++;
    }
  bool ok = a[21] > 0.5 && chol_inverse6(AtA, cov);
  if (ok)
    for (int i = 0; i < 36; i++) ok = ok && isfinite(cov[i]);
  for (int i = 0; i < 36; i++) out36[i] = ok ? cov[i] : ((i % 7 == 0) ? 1e6 : 0.0);
}

__device__ __forceinline__ void k_cov_prepare_body(IcpDeviceState* __restrict__ st, const SolveK* __restrict__ kp, uint32_t force) {
  if (!force && (!st->done || st->cov_done)) return;
  const int j = threadIdx.x;
  if (j >= 6) return;
  Pose Tc;
  for (int i = 0; i < 12; i++) Tc.m[i] = st->T[i];
  double out[12];
  cov_prepare_lane(Tc, j, kp->cov_hx, kp->cov_ha, out);
  for (int i = 0; i < 12; i++) st->covD[j * 12 + i] = out[i];
}

__device__ __forceinline__ void k_cov_accum_body(const IcpDeviceState* __restrict__ st, uint32_t force,
                                                      const float* __restrict__ lx, const float* __restrict__ ly,
                                                      const float* __restrict__ lz, uint32_t n,
                                                      const uint32_t* __restrict__ pair_gidx,
                                                      double* __restrict__ partials, uint32_t pstride) {
  __shared__ double sD[72];
  __shared__ BlockSum<kCovN> lds;
  if (!force && (!st->done || st->cov_done)) return;
  if (threadIdx.x < 72) sD[threadIdx.x] = st->covD[threadIdx.x];
  __syncthreads();
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  double v[kCovN];
#pragma unroll
  for (int j = 0; j < kCovN; j++) v[j] = 0.0;
  if (i < n && pair_gidx[i] != kNoMatch) cov_rows_point(sD, lx[i], ly[i], lz[i], v);
  block_sum_rows<kCovN>(v, lds, partials, pstride, blockIdx.x);
}

__global__ __launch_bounds__(kBlock) void k_cov_accum_pl(const IcpDeviceState* __restrict__ st,
                                                         const float* __restrict__ l3, const float* __restrict__ n3,
                                                         uint32_t n, uint32_t stride, double* __restrict__ partials,
                                                         uint32_t pstride) {
  __shared__ double sD[72];
  __shared__ BlockSum<kCovN> lds;
  if (threadIdx.x < 72) sD[threadIdx.x] = st->covD[threadIdx.x];
  __syncthreads();
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  double v[kCovN];
#pragma unroll
  for (int j = 0; j < kCovN; j++) v[j] = 0.0;
  if (i < n) {
    const double x = l3[i], y = l3[stride + i], z = l3[2 * stride + i];
    const double nx = n3[i], ny = n3[stride + i], nz = n3[2 * stride + i];
    double A[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      double r[3];
#pragma unroll
      for (int q = 0; q < 3; q++)
        r[q] = sD[j * 12 + q * 4] * x + sD[j * 12 + q * 4 + 1] * y + sD[j * 12 + q * 4 + 2] * z + sD[j * 12 + q * 4 + 3];
      A[j] = nx * r[0] + ny * r[1] + nz * r[2];
    }
    int q = 0;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int b = a; b < 6; b++) v[q++] = A[a] * A[b];
    v[21] = 1.0;
  }
  block_sum_rows<kCovN>(v, lds, partials, pstride, blockIdx.x);
}

// covariance rows of the stored point-to-plane pairings (fused path)
__device__ __forceinline__ void k_cov_accum_plbuf_body(const IcpDeviceState* __restrict__ st,
                                                       const float* __restrict__ lx, const float* __restrict__ ly,
                                                       const float* __restrict__ lz, uint32_t n,
                                                       const float4* __restrict__ pl_c, const float4* __restrict__ pl_n,
                                                       double* __restrict__ partials, uint32_t pstride) {
  __shared__ double sD[72];
  __shared__ BlockSum<kCovN> lds;
  if (!st->done || st->cov_done) return;
  if (threadIdx.x < 72) sD[threadIdx.x] = st->covD[threadIdx.x];
  __syncthreads();
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  double v[kCovN];
#pragma unroll
  for (int j = 0; j < kCovN; j++) v[j] = 0.0;
  if (i < n && pl_c[i].w != 0.f) {
    const float4 nn = pl_n[i];
    cov_rows_plane(sD, lx[i], ly[i], lz[i], (double)nn.x, (double)nn.y, (double)nn.z, v);
  }
  block_sum_rows<kCovN>(v, lds, partials, pstride, blockIdx.x);
}
__global__ __launch_bounds__(kBlock) void k_cov_accum_plbuf(const IcpDeviceState* __restrict__ st,
                                                            const float* __restrict__ lx, const float* __restrict__ ly,
                                                            const float* __restrict__ lz, uint32_t n,
                                                            const float4* __restrict__ pl_c, const float4* __restrict__ pl_n,
                                                            double* __restrict__ partials, uint32_t pstride) {
  k_cov_accum_plbuf_body(st, lx, ly, lz, n, pl_c, pl_n, partials, pstride);
}

__device__ __forceinline__ void k_cov_finalize_body(IcpDeviceState* __restrict__ st, uint32_t force,
                                                                const double* __restrict__ partA, uint32_t nA,
                                                                uint32_t strideA, const double* __restrict__ partB,
                                                                uint32_t nB, uint32_t strideB) {
  __shared__ double red[kCovN][64];
  __shared__ double totA[kCovN], totB[kCovN];
  if (!force && (!st->done || st->cov_done)) return;
  const int lane = threadIdx.x;
  if (nA) reduce_rows(partA, nA, strideA, kCovN, totA, red);
  if (nB) reduce_rows(partB, nB, strideB, kCovN, totB, red);
  if (lane != 0) return;
  double a[kCovN];
#pragma unroll
  for (int i = 0; i < kCovN; i++) a[i] = (nA ? totA[i] : 0.0) + (nB ? totB[i] : 0.0);
  double cov[36];
  cov_from_sums(a, cov);
  for (int i = 0; i < 36; i++) st->cov[i] = cov[i];
  st->cov_done = 1;
}

// ================================================================================================
// Pairing compaction: per-block count -> scan of block counts -> ballot/prefix scatter.
// Output order = ascending local index (what a serial matcher emits).
// ---- kernel entry points of the bodies above: one alignment per launch, or one job per blockIdx.y -------------------
__global__ __launch_bounds__(kBlock, MH_QUAD_WAVES) void k_match4(const IcpDeviceState* __restrict__ st,
                                                   const float* __restrict__ lx, const float* __restrict__ ly,
                                                   const float* __restrict__ lz, uint32_t n, MapView map,
                                                   float4* __restrict__ pair_q, uint32_t* __restrict__ pair_gidx,
                                                   const uint32_t* __restrict__ perm
#ifdef MH_DEBUG_WAVETRACE
                                                   , unsigned long long* __restrict__ wtrace
#endif
) {
  k_match4_body(st, lx, ly, lz, n, map, pair_q, pair_gidx, perm
#ifdef MH_DEBUG_WAVETRACE
                , wtrace
#endif
  );
}
// ================================================================================================
// k_match_flat: the plan / scan matcher (mh_nn_flat.h) -- a wave per 64 consecutive scan points, per-point, per-voxel and
// per-record work each spread over all 64 lanes and handed on through the wave's slice of LDS.  Device-state driven like
// k_match4 (same arguments, same pairings, bit for bit); the first Gauss-Newton accumulation is the k_accum launch that follows.
// ================================================================================================
#ifndef MH_FLAT_WAVES
#define MH_FLAT_WAVES 6  // waves per SIMD the register allocator has to leave room for (80 VGPRs); the LDS allows 5.5
#endif
#ifndef MH_FLAT_THREADS
#define MH_FLAT_THREADS 64
#endif
constexpr uint32_t kFlatThreads = MH_FLAT_THREADS;  // ONE wave per workgroup (nothing is shared between waves): 0.1637 ms per launch against 0.1787 with two and 0.1838 with four -- a workgroup holds its slot until its slowest wave is done
constexpr uint32_t kFlatPointsPerBlock = kFlatThreads; // a lane per point in phase A
__device__ __forceinline__ uint32_t nblk_flat_dev(uint32_t n) { return (n + kFlatPointsPerBlock - 1u) / kFlatPointsPerBlock; }
// (the grid width is a multiple of 8 = the XCDs a launch is dealt over, whatever the layer's size)
constexpr uint32_t kFlatGridUnit = 8u;
inline uint32_t nblk_flat(size_t n) { return (uint32_t)(((n + kFlatPointsPerBlock - 1) / kFlatPointsPerBlock + kFlatGridUnit - 1) / kFlatGridUnit * kFlatGridUnit); }
__device__ __forceinline__ void k_match_flat_body(const IcpDeviceState* __restrict__ st, const float* __restrict__ lx,
                                                  const float* __restrict__ ly, const float* __restrict__ lz, uint32_t n,
                                                  MapView map, float4* __restrict__ pair_q, uint32_t* __restrict__ pair_gidx,
                                                  const uint32_t* __restrict__ perm, uint32_t block_x) {
  __shared__ FlatWave sh[kFlatThreads / 64];
  typedef const IcpDeviceState __attribute__((address_space(4))) * cstate_ptr;
  const cstate_ptr cst = (cstate_ptr)uniform_const_ptr(st);
  if (cst->done) return;  // grid-uniform
  const uint32_t bx = block_x;
  const uint32_t i0 = bx * kFlatPointsPerBlock + (threadIdx.x & ~63u);
  if (i0 >= n) return;    // whole waves
  const bool have_prev = cst->iter > 0 && !map.no_prev_bound;
  double T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = cst->T[k];
  match_flat_wave(sh[threadIdx.x >> 6], map, T, cst->cur_thr2, cst->cur_ang2, have_prev, lx, ly, lz, n, i0, pair_q, pair_gidx, perm);
}
__global__ __launch_bounds__(kFlatThreads, MH_FLAT_WAVES) void k_match_flat(const IcpDeviceState* __restrict__ st, const float* __restrict__ lx,
                                                             const float* __restrict__ ly, const float* __restrict__ lz, uint32_t n,
                                                             MapView map, float4* __restrict__ pair_q,
                                                             uint32_t* __restrict__ pair_gidx, const uint32_t* __restrict__ perm) {
  k_match_flat_body(st, lx, ly, lz, n, map, pair_q, pair_gidx, perm, blockIdx.x);
}
__global__ __launch_bounds__(kBlock, MH_QUAD_WAVES) void k_match4_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  k_match4_body(j.st, j.lx, j.ly, j.lz, j.n, j.map, j.pair_q, j.pair_gidx, nullptr
#ifdef MH_DEBUG_WAVETRACE
                , nullptr
#endif
  );
}
// A whole job per XCD.  Workgroup L of a launch runs on XCD L % 8 (tools/xcd_exchange.hip: 0 exceptions in 256; the grid's width is
// a multiple of 8), so with blockIdx.y = job every job's workgroups are dealt over all eight XCDs and every XCD's L2 fetches its own
// copy of every job's map: (2 FETCH_SIZE + WRITE_SIZE) = 1.64 x the compulsory bytes on C2.  Here XCD c takes the jobs c, c + 8, ...
// of the first 8 * floor(jobs / 8) one after the other (the rest keep the plain order): a map is fetched into ONE L2 --
// FETCH_SIZE 155.5 -> 89.7 MB per launch of 32 scans, traffic 1.64 -> 1.08 x compulsory, L2 hit rate 47 -> 69 %, headline
// 6320-6330 -> 6470-6480 scans/s.  (Unlike a contiguous PART of one scan per XCD -- profiles/r05_match_kernel.md section 5: 1.22 x
// at -37 % speed -- whole scans are equal work.)  -DMH_FLAT_NO_JOB_XCD: the plain order (A/B).
__global__ __launch_bounds__(kFlatThreads, MH_FLAT_WAVES) void k_match_flat_b(const BatchJob* __restrict__ jobs) {
  uint32_t job = blockIdx.y, bx = blockIdx.x;
  const uint32_t whole = gridDim.y & ~7u;  // jobs that are dealt an XCD each
  if (blockIdx.y < whole) {
    const uint32_t L = blockIdx.x + blockIdx.y * gridDim.x, xcd = L % 8u, slot = L / 8u;
    job = xcd + 8u * (slot / gridDim.x);
    bx = slot % gridDim.x;
  }
  const BatchJob& j = jobs[job];
  k_match_flat_body(j.st, j.lx, j.ly, j.lz, j.n, j.map, j.pair_q, j.pair_gidx, nullptr, bx);
}
#ifdef MH_DEV_VARIANTS
#include "mh_dev_variants.h"  // k_match_tile*, k_match_wave_*, k_match4o_b: development library only
#endif
template <bool SIGNED>
__global__ __launch_bounds__(kBlock, MH_ACCUM_WAVES) void k_accum(const IcpDeviceState* __restrict__ st, uint32_t first,
                                                  const MatchK* __restrict__ kp, const float* __restrict__ lx,
                                                  const float* __restrict__ ly, const float* __restrict__ lz, uint32_t n,
                                                  const float4* __restrict__ pair_q,
                                                  const uint32_t* __restrict__ pair_gidx, double* __restrict__ partials,
                                                  uint32_t pstride) {
  k_accum_body<SIGNED>(st, first, kp, lx, ly, lz, n, pair_q, pair_gidx, partials, pstride, blockIdx.x);
}
template <bool SIGNED>
__global__ __launch_bounds__(kBlock, MH_ACCUM_WAVES) void k_accum_b(const BatchJob* __restrict__ jobs, uint32_t first) {
  // (the matcher's job -> XCD mapping was tried here as well -- the pairings this launch reads were written through that XCD's L2 --
  //  and changes nothing: 22.4-22.9 us either way; a launch boundary leaves nothing of them in the L2)
  const BatchJob& j = jobs[blockIdx.y];
  const uint32_t bx = blockIdx.x;
  if (bx >= j.nba) return;
  k_accum_body<SIGNED>(j.st, first, j.mk, j.lx, j.ly, j.lz, j.n, j.pair_q, j.pair_gidx, j.part, j.nba, bx);
}
__global__ __launch_bounds__(kSolveThreads) void k_solve(IcpDeviceState* __restrict__ st, const SolveK* __restrict__ kp,
                                                         const double* __restrict__ partA, uint32_t nA, uint32_t strideA,
                                                         const double* __restrict__ partB, uint32_t nB,
                                                         uint32_t strideB, uint32_t first) {
  k_solve_body(st, kp, partA, nA, strideA, partB, nB, strideB, first);
}
__global__ __launch_bounds__(kSolveThreads) void k_solve_b(const BatchJob* __restrict__ jobs, uint32_t first) {
  const BatchJob& j = jobs[blockIdx.y];
  const uint32_t cols = first ? j.nbm : j.nba;  // the first step's partials come from the matcher-side producer
  k_solve_body(j.st, j.sk, j.part, cols, cols, nullptr, 0u, 0u, first);
}
__global__ void k_cov_prepare(IcpDeviceState* __restrict__ st, const SolveK* __restrict__ kp, uint32_t force) {
  k_cov_prepare_body(st, kp, force);
}
__global__ void k_cov_prepare_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  k_cov_prepare_body(j.st, j.sk, 0u);
}
__global__ __launch_bounds__(kBlock) void k_cov_accum(const IcpDeviceState* __restrict__ st, uint32_t force,
                                                      const float* __restrict__ lx, const float* __restrict__ ly,
                                                      const float* __restrict__ lz, uint32_t n,
                                                      const uint32_t* __restrict__ pair_gidx,
                                                      double* __restrict__ partials, uint32_t pstride) {
  k_cov_accum_body(st, force, lx, ly, lz, n, pair_gidx, partials, pstride);
}
__global__ __launch_bounds__(kBlock) void k_cov_accum_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  if (blockIdx.x >= j.nb) return;
  k_cov_accum_body(j.st, 0u, j.lx, j.ly, j.lz, j.n, j.pair_gidx, j.part, j.nb);
}
__global__ __launch_bounds__(kSolveThreads) void k_cov_finalize(IcpDeviceState* __restrict__ st, uint32_t force,
                                                                const double* __restrict__ partA, uint32_t nA,
                                                                uint32_t strideA, const double* __restrict__ partB,
                                                                uint32_t nB, uint32_t strideB) {
  k_cov_finalize_body(st, force, partA, nA, strideA, partB, nB, strideB);
}
__global__ __launch_bounds__(kSolveThreads) void k_cov_finalize_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  k_cov_finalize_body(j.st, 0u, j.part, j.nb, j.nb, j.partb, j.partb ? j.nb : 0u, j.partb ? j.nb : 0u);
}
__global__ __launch_bounds__(kBlock) void k_cov_accum_plbuf_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  if (blockIdx.x >= j.nb || !j.partb) return;
  k_cov_accum_plbuf_body(j.st, j.lx, j.ly, j.lz, j.n, j.pl_c, j.pl_n, j.partb, j.nb);
}
template <bool PL, bool FUSED>
__global__ __launch_bounds__(kBlock) void k_match16(const IcpDeviceState* __restrict__ st, const MatchK* __restrict__ kp,
                                                    const float* __restrict__ lx, const float* __restrict__ ly,
                                                    const float* __restrict__ lz, uint32_t n, MapView map,
                                                    float4* __restrict__ pair_q, uint32_t* __restrict__ pair_gidx,
                                                    float4* __restrict__ pl_c, float4* __restrict__ pl_n,
                                                    double* __restrict__ partials, uint32_t pstride) {
  k_match16_body<PL, FUSED>(st, kp, lx, ly, lz, n, map, pair_q, pair_gidx, pl_c, pl_n, partials, pstride);
}
// row kernel with the fused first accumulation, one job per blockIdx.y (layers of 2-12 k points in lock step)
__global__ __launch_bounds__(kBlock) void k_match16f_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  if (blockIdx.x >= j.nbm) return;
  k_match16_body<false, true>(j.st, j.mk, j.lx, j.ly, j.lz, j.n, j.map, j.pair_q, j.pair_gidx, nullptr, nullptr, j.part, j.nbm);
}
// start of a lock-step batch: the staged [state | params | schedules] of all jobs -> where each job keeps them
__global__ void k_scatter_blocks(const BatchJob* __restrict__ jobs, const uint32_t* __restrict__ stage, uint32_t dwords) {
  const BatchJob& j = jobs[blockIdx.x];
  uint32_t* dst = reinterpret_cast<uint32_t*>(j.st);
  const uint32_t* src = stage + j.stage_off;
  for (uint32_t i = threadIdx.x; i < dwords; i += blockDim.x) dst[i] = src[i];
  for (uint32_t i = threadIdx.x; i < j.sched_dwords; i += blockDim.x) j.sched_dst[i] = src[dwords + i];
}
// all jobs' state blocks into one contiguous buffer: one read-back per chunk instead of one per job
__global__ void k_gather_states(const BatchJob* __restrict__ jobs, IcpDeviceState* __restrict__ out) {
  const uint32_t* src = reinterpret_cast<const uint32_t*>(jobs[blockIdx.x].st);
  uint32_t* dst = reinterpret_cast<uint32_t*>(out + blockIdx.x);
  for (uint32_t i = threadIdx.x; i < sizeof(IcpDeviceState) / 4; i += blockDim.x) dst[i] = src[i];
}

// ================================================================================================
__device__ __forceinline__ void k_count_valid_body(const uint32_t* __restrict__ gidx, uint32_t n,
                                                   uint32_t* __restrict__ block_counts) {
  __shared__ uint32_t wc[kBlock / 64];
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const bool v = i < n && gidx[i] != kNoMatch;
  const unsigned long long m = __ballot(v);
  if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}
__global__ __launch_bounds__(kBlock) void k_count_valid(const uint32_t* __restrict__ gidx, uint32_t n,
                                                        uint32_t* __restrict__ block_counts) {
  k_count_valid_body(gidx, n, block_counts);
}

__device__ __forceinline__ void k_scan_blocks_body(const uint32_t* __restrict__ counts, uint32_t nb,
                                                   uint32_t* __restrict__ offsets, uint32_t* __restrict__ total) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t base = 0; base < nb; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t c = i < nb ? counts[i] : 0;
    uint32_t incl = c;  // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up((int)incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t wpre = 0;
    for (int w = 0; w < wave; w++) wpre += wsum[w];
    if (i < nb) offsets[i] = carry + wpre + incl - c;
    __syncthreads();
    if (threadIdx.x == 1023) carry += wpre + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total) *total = carry;
}
__global__ __launch_bounds__(1024) void k_scan_blocks(const uint32_t* __restrict__ counts, uint32_t nb,
                                                      uint32_t* __restrict__ offsets, uint32_t* __restrict__ total) {
  k_scan_blocks_body(counts, nb, offsets, total);
}

__device__ __forceinline__ void k_compact_body(const uint32_t* __restrict__ gidx, const float4* __restrict__ pq,
                                               uint32_t n, const uint32_t* __restrict__ block_offsets,
                                               uint32_t* __restrict__ o_li, uint32_t* __restrict__ o_gi,
                                               float* __restrict__ o_x, float* __restrict__ o_y,
                                               float* __restrict__ o_z, float* __restrict__ o_d2) {
  __shared__ uint32_t wc[kBlock / 64];
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const uint32_t gi = i < n ? gidx[i] : kNoMatch;
  const bool v = gi != kNoMatch;
  const unsigned long long m = __ballot(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wc[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (!v) return;
  uint32_t pos = block_offsets[blockIdx.x] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; w++) pos += wc[w];
  const float4 q = pq[i];
  if (o_li) o_li[pos] = i;
  if (o_gi) o_gi[pos] = gi;
  if (o_x) o_x[pos] = q.x;
  if (o_y) o_y[pos] = q.y;
  if (o_z) o_z[pos] = q.z;
  if (o_d2) o_d2[pos] = q.w;
}
__global__ __launch_bounds__(kBlock) void k_compact(const uint32_t* __restrict__ gidx, const float4* __restrict__ pq,
                                                    uint32_t n, const uint32_t* __restrict__ block_offsets,
                                                    uint32_t* __restrict__ o_li, uint32_t* __restrict__ o_gi,
                                                    float* __restrict__ o_x, float* __restrict__ o_y,
                                                    float* __restrict__ o_z, float* __restrict__ o_d2) {
  k_compact_body(gidx, pq, n, block_offsets, o_li, o_gi, o_x, o_y, o_z, o_d2);
}
// the same three steps for every job of a batch (blockIdx.y = job), once the job's loop has terminated: the final
// pairings of job j land in its part of the batch's pairs block, ascending local index
__global__ __launch_bounds__(kBlock) void k_count_valid_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  if (blockIdx.x >= j.nb || !j.cp_out || !j.st->done) return;
  k_count_valid_body(j.pair_gidx, j.n, j.cp_counts);
}
__global__ __launch_bounds__(1024) void k_scan_blocks_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  if (!j.cp_out || !j.st->done) return;
  k_scan_blocks_body(j.cp_counts, j.nb, j.cp_counts + j.nb, nullptr);
}
__global__ __launch_bounds__(kBlock) void k_compact_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  if (blockIdx.x >= j.nb || !j.cp_out || !j.st->done) return;
  const uint32_t S = j.cp_stride;
  k_compact_body(j.pair_gidx, j.pair_q, j.n, j.cp_counts + j.nb, j.cp_out, j.cp_out + S, (float*)(j.cp_out + 2 * S),
                 (float*)(j.cp_out + 3 * S), (float*)(j.cp_out + 4 * S), (float*)(j.cp_out + 5 * S));
}

// point-to-plane pairings: flags + compaction in ascending local index
__global__ void k_pl_flags(const float4* __restrict__ pl_c, uint32_t n, uint32_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = pl_c[i].w != 0.f ? i : kNoMatch;
}
__global__ __launch_bounds__(kBlock) void k_compact_pl(const uint32_t* __restrict__ flags, const float4* __restrict__ pl_c,
                                                       const float4* __restrict__ pl_n, uint32_t n,
                                                       const uint32_t* __restrict__ block_offsets, uint32_t* __restrict__ o_li,
                                                       float* __restrict__ o_cx, float* __restrict__ o_cy,
                                                       float* __restrict__ o_cz, float* __restrict__ o_nx,
                                                       float* __restrict__ o_ny, float* __restrict__ o_nz) {
  __shared__ uint32_t wc[kBlock / 64];
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const bool v = i < n && flags[i] != kNoMatch;
  const unsigned long long m = __ballot(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wc[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (!v) return;
  uint32_t pos = block_offsets[blockIdx.x] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; w++) pos += wc[w];
  const float4 c = pl_c[i], nn = pl_n[i];
  if (o_li) o_li[pos] = i;
  if (o_cx) o_cx[pos] = c.x;
  if (o_cy) o_cy[pos] = c.y;
  if (o_cz) o_cz[pos] = c.z;
  if (o_nx) o_nx[pos] = nn.x;
  if (o_ny) o_ny[pos] = nn.y;
  if (o_nz) o_nz[pos] = nn.z;
}

// dense outputs of the un-compacted search
__global__ void k_unpack_dense(const uint32_t* __restrict__ gidx, const float4* __restrict__ pq, uint32_t n,
                               uint32_t* __restrict__ o_gi, float* __restrict__ o_x, float* __restrict__ o_y,
                               float* __restrict__ o_z, float* __restrict__ o_d2) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 q = pq[i];
  if (o_gi) o_gi[i] = gidx[i];
  if (o_x) o_x[i] = q.x;
  if (o_y) o_y[i] = q.y;
  if (o_z) o_z[i] = q.z;
  if (o_d2) o_d2[i] = q.w;
}

// solver-granular path: pack caller pairings into the pair buffers
__global__ void k_pack_pairs(const float* __restrict__ g3, uint32_t n, uint32_t stride, float4* __restrict__ pq,
                             uint32_t* __restrict__ gidx) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  pq[i] = make_float4(g3[i], g3[stride + i], g3[2 * stride + i], 0.f);
  gidx[i] = i;
}

// ================================================================================================
// Host side
// ================================================================================================
namespace {

inline uint32_t nblk(size_t n) { return (uint32_t)((n + kBlock - 1) / kBlock); }


// One device block [state | parameters] with a pinned mirror of the same layout: an alignment starts with ONE upload.
constexpr size_t kParamsOffset = (sizeof(IcpDeviceState) + 255) / 256 * 256;
// The threshold schedules of a single alignment (threshold | kernel_param | pt2pl_threshold, max_iterations doubles each) ride
// behind the parameters in the same block when they fit: state, parameters and schedules go up in ONE copy instead of two.
constexpr size_t kInlineSchedDoubles = 3 * 400;
constexpr size_t kInlineSchedOffset = (kParamsOffset + sizeof(IcpDeviceParams) + 63) / 64 * 64;

mh_status ensure_state(mh_ctx* ctx) {
  if (!ctx->d_state) {
    char *d = nullptr, *h = nullptr;
    const size_t second_off = (kInlineSchedOffset + kInlineSchedDoubles * sizeof(double) + 255) / 256 * 256;  // k_step16's two ping-pong state blocks (heads only)
    MH_HIP(hipMalloc((void**)&d, second_off + 2 * 256));
    static_assert(kStateHeadDwords * 4 <= 256, "a ping-pong block per 256 bytes");
    const size_t prog_off = ((kInlineSchedOffset + kInlineSchedDoubles * sizeof(double) + 127) / 128) * 128;  // a cache line of its own
    MH_HIP(hipHostMalloc((void**)&h, prog_off + 128, hipHostMallocDefault));
    ctx->d_state = (IcpDeviceState*)d;
    ctx->d_state_b = (IcpDeviceState*)(d + second_off);
    ctx->h_state = (IcpDeviceState*)h;
    ctx->d_params = (IcpDeviceParams*)(d + kParamsOffset);
    ctx->h_params = (IcpDeviceParams*)(h + kParamsOffset);
    // the word the device loop publishes its progress in (run_streaming): page-locked host memory, written by the kernels
    ctx->h_progress = (uint32_t*)(h + prog_off);
    *ctx->h_progress = 0;
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, ctx->h_progress, 0) != hipSuccess) dp = nullptr;  // (then the loop is never streamed)
    ctx->d_progress = (uint32_t*)dp;
  }
  return MH_OK;
}

// state + parameters (+ the schedules that ride behind them: inline_sched doubles) in one copy (start of mh_icp_align)
mh_status upload_state_and_params(mh_ctx* ctx, const MatchK& mk, const SolveK& sk, size_t inline_sched = 0, hipStream_t on = nullptr) {
  ctx->h_params->mk = mk;
  ctx->h_params->sk = sk;
  const size_t bytes = inline_sched ? kInlineSchedOffset + inline_sched * sizeof(double) : kParamsOffset + sizeof(IcpDeviceParams);
  MH_HIP(hipMemcpyAsync(ctx->d_state, ctx->h_state, bytes, hipMemcpyHostToDevice, on ? on : ctx->stream));
  return MH_OK;
}
// the context's loop stream (created on first use), ordered behind everything queued on its own stream so far
mh_status loop_stream_behind(mh_ctx* ctx, hipStream_t* out) {
  if (!ctx->loop_stream) {
    int lo = 0, hi = 0;  // (numerically lower = higher priority)
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = hi = 0;
    MH_HIP(hipStreamCreateWithPriority(&ctx->loop_stream, hipStreamNonBlocking, hi));
    MH_HIP(hipEventCreateWithFlags(&ctx->ev_loop, hipEventDisableTiming));
  }
  MH_HIP(hipEventRecord(ctx->ev_loop, ctx->stream));
  MH_HIP(hipStreamWaitEvent(ctx->loop_stream, ctx->ev_loop, 0));
  *out = ctx->loop_stream;
  return MH_OK;
}

// parameters -> pinned mirror -> device block (the caller has synchronised the previous use of the mirror)
mh_status upload_params(mh_ctx* ctx, const MatchK& mk, const SolveK& sk) {
  ctx->h_params->mk = mk;
  ctx->h_params->sk = sk;
  MH_HIP(hipMemcpyAsync(ctx->d_params, ctx->h_params, sizeof(IcpDeviceParams), hipMemcpyHostToDevice, ctx->stream));
  return MH_OK;
}

mh_status ensure_pair_buffers(mh_ctx* ctx, size_t n) {
  const size_t nn = n ? n : 1;
  MH_TRY(ctx->pair_q.reserve(nn * sizeof(float4)));
  MH_TRY(ctx->pair_gidx.reserve(nn * sizeof(uint32_t)));
  const size_t nb = nblk(nn);
  MH_TRY(ctx->partials.reserve((size_t)kGenN * nb * sizeof(double)));
  return MH_OK;
}

mh_status ensure_pl_buffers(mh_ctx* ctx, size_t n) {
  const size_t nn = n ? n : 1;
  MH_TRY(ctx->pl_c.reserve(nn * sizeof(float4)));
  MH_TRY(ctx->pl_n.reserve(nn * sizeof(float4)));
  MH_TRY(ctx->partials_b.reserve((size_t)kGenN * nblk(nn) * sizeof(double)));
  return MH_OK;
}

void init_state(IcpDeviceState* h, const double T[12]) {
  memset(h, 0, sizeof(*h));
  for (int i = 0; i < 12; i++) {
    h->T[i] = T[i];
    h->T_prev[i] = T[i];
  }
  h->solver_ok = 1;
  for (int i = 0; i < 6; i++) h->cov[i * 7] = 1e6;
}

void fill_prior(SolveK& k, const mh_prior* prior) {
  k.has_prior = prior ? 1u : 0u;
  if (!prior) return;
  Pose P;
  for (int i = 0; i < 12; i++) P.m[i] = prior->mean[i];
  const Pose Pi = inverse(P);
  for (int i = 0; i < 12; i++) k.prior_mean_inv[i] = Pi.m[i];
  for (int i = 0; i < 36; i++) k.prior_info[i] = prior->info[i];
}

bool pose_ok(const double T[12]) {
  for (int i = 0; i < 12; i++)
    if (!isfinite(T[i])) return false;
  return true;
}

// compaction of the context's pair buffers into caller arrays; returns the number of pairs
mh_status compact_pairs(mh_ctx* ctx, size_t n, const mh_pairs_out* out, int32_t mem, uint64_t* n_pairs_out) {
  hipStream_t s = ctx->stream;
  const uint32_t nb = nblk(n);
  const size_t n4 = ((n + 63) / 64) * 64;
  // layout: counts[nb] | offsets[nb] | total[1] | (host staging) li,gi,x,y,z,d2 [n4 each]
  const size_t hdr = (((size_t)2 * nb + 1) * 4 + 255) / 256 * 256;
  MH_TRY(ctx->compact.reserve(hdr + 6 * n4 * 4));
  uint32_t* counts = ctx->compact.as<uint32_t>();
  uint32_t* offsets = counts + nb;
  uint32_t* total = offsets + nb;
  char* stage = ctx->compact.as<char>() + hdr;
  uint32_t *o_li, *o_gi;
  float *o_x, *o_y, *o_z, *o_d2;
  if (mem == MH_MEM_DEVICE) {
    o_li = out->local_idx; o_gi = out->global_idx; o_x = out->gx; o_y = out->gy; o_z = out->gz; o_d2 = out->d2;
  } else {
    o_li = out->local_idx ? (uint32_t*)(stage) : nullptr;
    o_gi = out->global_idx ? (uint32_t*)(stage + n4 * 4) : nullptr;
    o_x = out->gx ? (float*)(stage + 2 * n4 * 4) : nullptr;
    o_y = out->gy ? (float*)(stage + 3 * n4 * 4) : nullptr;
    o_z = out->gz ? (float*)(stage + 4 * n4 * 4) : nullptr;
    o_d2 = out->d2 ? (float*)(stage + 5 * n4 * 4) : nullptr;
  }
  uint32_t h_total = 0;
  if (n) {
    hipLaunchKernelGGL(k_count_valid, dim3(nb), dim3(kBlock), 0, s, ctx->pair_gidx.as<uint32_t>(), (uint32_t)n, counts);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, s, counts, nb, offsets, total);
    hipLaunchKernelGGL(k_compact, dim3(nb), dim3(kBlock), 0, s, ctx->pair_gidx.as<uint32_t>(), ctx->pair_q.as<float4>(),
                       (uint32_t)n, offsets, o_li, o_gi, o_x, o_y, o_z, o_d2);
    MH_HIP(hipGetLastError());
    MH_HIP(hipMemcpyAsync(&h_total, total, 4, hipMemcpyDeviceToHost, s));
    MH_HIP(mh::wait_stream(s));
  }
  if (mem == MH_MEM_HOST && h_total) {
    const size_t b = (size_t)h_total * 4;
    if (out->local_idx) MH_HIP(hipMemcpy(out->local_idx, o_li, b, hipMemcpyDeviceToHost));
    if (out->global_idx) MH_HIP(hipMemcpy(out->global_idx, o_gi, b, hipMemcpyDeviceToHost));
    if (out->gx) MH_HIP(hipMemcpy(out->gx, o_x, b, hipMemcpyDeviceToHost));
    if (out->gy) MH_HIP(hipMemcpy(out->gy, o_y, b, hipMemcpyDeviceToHost));
    if (out->gz) MH_HIP(hipMemcpy(out->gz, o_z, b, hipMemcpyDeviceToHost));
    if (out->d2) MH_HIP(hipMemcpy(out->d2, o_d2, b, hipMemcpyDeviceToHost));
  }
  if (n_pairs_out) *n_pairs_out = h_total;
  return MH_OK;
}

// One alignment in flight on one context: enqueue / poll state machine shared by mh_icp_align and
// mh_icp_align_batch.
std::atomic<unsigned long long> g_loop16_runs{0}, g_loop16_fallbacks{0};  // one-launch loops started / abandoned for the chain (mh_debug_loop_stats)
uint32_t loop_cu_limit(int device) {
  static uint32_t cap[64] = {0};
  uint32_t& cu = cap[(unsigned)device % 64u];
  if (!cu) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || v <= 0) v = 64;
    cu = (uint32_t)v;
  }
  const char* e = getenv("MH_LOOP16_CUS");  // (test knob: the admission limit)
  return e ? (uint32_t)std::max(0, atoi(e)) : cu;
}
// the loops of `callers` alignments of `ng` groups each fit 70 % of the device's CUs side by side (one group per workgroup; shared
// loops whose workgroups take 2 or 4 groups each were measured for 6 / 8 / 16 callers and lost to lock-step batches: 3100-3550 /
// 3260-3540 / 3000-3140 against 4960 / 5300 / 5470-6150 scans/s -- the callers' other kernels start and stop beside them)
// Admission is counted in HALF compute units: a k_icp16 workgroup (512 lanes x ~250 registers) takes a CU, a k_icpw workgroup
// (256 lanes, 63 KB of LDS) half of one.
constexpr uint32_t kLoopUnitsPerCu = 2;
// MH_LOOPW = all | solo | batch | none: where k_icpw / k_icpw_b (plan / scan search, 128 points per workgroup) replace k_icp16 /
// k_icp16_b (a DPP row per point) on point layers.  (MH_NO_LOOPW=1 = none.)
bool loop_wave_enabled(bool batch = false) {
  if (getenv("MH_NO_LOOPW") != nullptr) return false;
  const char* e = getenv("MH_LOOPW");
  if (!e) e = MH_LOOPW_DEFAULT;
  return e[0] == 'a' || (batch ? e[0] == 'b' : e[0] == 's');
}
// what the loop of a layer of `ng` groups holds: with_planes -> k_icp16<true> (the plane matcher rides in the row search)
uint32_t loop_units(uint32_t ng, bool with_planes) {
  return (with_planes || !loop_wave_enabled()) ? kLoopUnitsPerCu * ng : (ng + kLwGroups - 1) / kLwGroups;
}
bool loops_fit(int device, uint32_t units, uint32_t callers) {
  // (and never more than MH_SOLO_MAX_CALLERS, 4: beyond, every measurement so far says lock-step batches -- the loops' own
  //  streams share four hardware queues with everything else the callers issue, and a loop holds its queue for 0.5 ms)
  const char* e = getenv("MH_SOLO_MAX_CALLERS");
  const uint32_t cap = e ? (uint32_t)std::max(1, atoi(e)) : 4u;
  return callers <= 1 || (callers <= cap && (uint64_t)callers * units * 10u <= (uint64_t)loop_cu_limit(device) * kLoopUnitsPerCu * 7u);
}

struct AlignJob {
  const mh_map* map = nullptr;
  const mh_scan* scan = nullptr;
  mh_ctx* ctx = nullptr;
  const mh_icp_params* p = nullptr;
  mh_icp_result* res = nullptr;
  mh_icp_iter* trace = nullptr;
  MatchK mk{};
  SolveK sk{};
  uint32_t nb = 0, nbm = 0, nba = 0, enqueued = 0, chunk = 0, prof_n = 0, polls = 0, kind = 0;
  uint32_t step_total = 0;  // ... launches of the chunks enqueued so far when they are replayed from a graph (step_launches only counts direct launches)
  uint32_t serial_base = 0, step_launches = 0;  // k_step16's hand-over: the state block's serial number at upload, launches since
  // k_step16 chain: the state block the next launch reads (2 = the canonical one: after the upload and after a close-only launch;
  // 0 / 1 = the ping-pong pair) and the half of the partials it reads
  uint32_t step_src = 2, step_ppar = 0;
  bool auto_chunk = false;
  bool fused16 = false;  // row kernel accumulates the first Gauss-Newton step itself (layers above the one-workgroup size)
  bool defer_upload = false;   // batches: the pinned mirrors are filled, the copies are issued by the batch (staged) or flush()
  size_t nsched_pending = 0;
  size_t inline_sched = 0;  // doubles of the schedules that ride behind the parameter block (single alignments; 0: their own copy)
  int variant = 0;
  bool finished = false, trivial = false;
  bool prof = false;  // time this job's match kernels with events (then it cannot use the graph path)
  bool pl = false;    // Matcher_Point2Plane runs before the point matcher (lidar3d-ndt.yaml:195-210)
  bool streaming = false;  // run_streaming(): iterations are enqueued one by one behind the device's published progress
  bool skip_tail = false;  // ... and the covariance kernels + state read-back only once the loop has ended
  bool loop16 = false;         // run_loop16(): the whole loop of a small layer in ONE launch (k_icp16 / k_icpw)
  hipStream_t ws = nullptr;    // the stream a one-launch loop, its upload and its tail run on (the context's loop stream; null: ctx->stream)
  bool loopw = false;          // ... as k_icpw (plan / scan search, 128 points per workgroup; needs the map's sub-voxel index)
  bool forbid_loop16 = false;  // ... not for this job: the second attempt after a loop whose workgroups gave up
  uint32_t loop_wgs = 0;       // workgroups this job holds of the device's admission count while its loop runs

  mh_status start(const mh_map* m, const mh_scan* sc, const mh_icp_params* prm, const double* T0, const mh_prior* prior,
                  mh_icp_result* r, mh_icp_iter* tr, size_t batch_index = 0) {
    map = m; scan = sc; ctx = sc->ctx; p = prm; res = r; trace = tr;
    prof = prm->profile == 1 || (prm->profile == 2 && batch_index == 0);
    pl = prm->pt2pl_threshold != nullptr;
    memset(res, 0, sizeof(*res));
    for (int i = 0; i < 12; i++) res->T[i] = T0[i];
    for (int i = 0; i < 6; i++) res->cov[i * 7] = 1e6;
    res->potential_pairings = scan->n * (pl ? 2u : 1u);  // every matcher adds its layer size (App.B U6)
    if (p->max_iterations == 0 || scan->n == 0) {
      // ICP::align with nothing to iterate on: no pairings, quality 0, cov = diag(1e6)
      res->termination_reason = p->max_iterations == 0 ? MH_TERM_MAX_ITERATIONS : MH_TERM_NO_PAIRINGS;
      finished = trivial = true;
      return MH_OK;
    }
    MH_TRY(set_device(ctx));
    MH_TRY(ensure_state(ctx));
    MH_TRY(ensure_pair_buffers(ctx, scan->n));
    hipStream_t s = ctx->stream;
    MH_TRY(map_ready_on(map, s));  // a key-frame update still running on the map's side stream (mh_map_insert)
    const size_t mi = p->max_iterations;
    if (pl) {
      if (!map->view().ndt)
        return fail(MH_ERR_INVALID_ARGUMENT, "pt2pl_threshold given but the map carries no NDT statistics "
                                             "(build it with ndt_max_eigen_ratio > 0)");
      MH_TRY(ensure_pl_buffers(ctx, scan->n));
    }
    MH_TRY(ctx->sched.reserve(3 * mi * sizeof(double)));
    {  // threshold | kernel_param | pt2pl_threshold schedules: packed in a pinned staging block, one upload
      const size_t nsched = (pl ? 3 : 2) * mi;
      if (ctx->h_sched_cap < nsched) {
        if (ctx->h_sched) (void)hipHostFree(ctx->h_sched);
        ctx->h_sched = nullptr;
        ctx->h_sched_cap = 0;
        MH_HIP(hipHostMalloc((void**)&ctx->h_sched, 3 * mi * sizeof(double), hipHostMallocDefault));
        ctx->h_sched_cap = 3 * mi;
      }
      memcpy(ctx->h_sched, p->threshold, mi * sizeof(double));
      memcpy(ctx->h_sched + mi, p->kernel_param, mi * sizeof(double));
      if (pl) {  // MH_PT2PL_CENTROID_DISTANCE travels as a negative threshold (pl_accept)
        const double sgn = p->pt2pl_mode == MH_PT2PL_CENTROID_DISTANCE ? -1.0 : 1.0;
        for (size_t k = 0; k < mi; k++) ctx->h_sched[2 * mi + k] = sgn * fabs(p->pt2pl_threshold[k]);
      }
      nsched_pending = nsched;
      // a single alignment whose schedules fit behind the parameter block: they travel with it (one copy, below)
      inline_sched = (!defer_upload && nsched <= kInlineSchedDoubles) ? nsched : 0;
      if (inline_sched)
        memcpy(reinterpret_cast<char*>(ctx->h_state) + kInlineSchedOffset, ctx->h_sched, nsched * sizeof(double));
      else if (!defer_upload)
        MH_HIP(hipMemcpyAsync(ctx->sched.p, ctx->h_sched, nsched * sizeof(double), hipMemcpyHostToDevice, s));
    }
    if (trace) MH_TRY(ctx->trace.reserve(mi * sizeof(mh_icp_iter)));
    ctx->align_serial++;
    init_state(ctx->h_state, T0);
    // k_step16's hand-over: the uploaded block carries this alignment's epoch, every launch one more (a launch told what to
    // expect waits for exactly that block: neither the previous alignment's nor the launch before last's will do)
    serial_base = ((uint32_t)ctx->align_serial & 0x3FFu) << 22;
    step_launches = 0;
    step_total = 0;
    ctx->h_state->serial = serial_base;
    const double ang = p->threshold_angular_deg * 3.14159265358979323846 / 180.0;
    ctx->h_state->cur_thr2 = (float)(p->threshold[0] * p->threshold[0]);
    ctx->h_state->cur_ang2 = (float)(ang * ang);
    ctx->h_state->cur_kparam = p->kernel_param[0];  // uploaded together with the parameters below

    double* const sched_dev = inline_sched ? reinterpret_cast<double*>(reinterpret_cast<char*>(ctx->d_state) + kInlineSchedOffset)
                                           : ctx->sched.as<double>();
    mk.thr = sched_dev;
    mk.kparam = sched_dev + mi;
    mk.ang2 = (float)(ang * ang);
    mk.kernel = p->gn.robust_kernel;
    mk.w_pt2pt = p->gn.weight_pt2pt;
    mk.w_pt2pl = p->gn.weight_pt2pl;
    mk.pl_thr = pl ? sched_dev + 2 * mi : nullptr;
    mk.skip_pl_paired = (pl && p->matched_points == MH_MATCHED_POINTS_SKIP) ? 1u : 0u;
    memset(&sk, 0, sizeof(sk));
    sk.max_iterations = p->max_iterations;
    sk.disable_stall = p->disable_stall_test;
    sk.max_inner = p->gn.max_inner_iterations;
    sk.min_step_trans = p->min_abs_step_trans;
    sk.min_step_rot = p->min_abs_step_rot;
    sk.min_delta = p->gn.min_delta;
    sk.max_cost = p->gn.max_cost;
    sk.hook_enabled = p->hook_enabled;
    sk.hook_trans = p->hook_min_trans;
    sk.hook_rot = p->hook_min_rot;
    if (p->hook_enabled) {
      Pose C;
      for (int i = 0; i < 12; i++) C.m[i] = p->hook_checkpoint[i];
      const Pose Ci = inverse(C);
      for (int i = 0; i < 12; i++) sk.hook_chk_inv[i] = Ci.m[i];
    }
    fill_prior(sk, prior);
    sk.thr = mk.thr;
    sk.kparam = mk.kparam;
    sk.trace = trace ? ctx->trace.as<mh_icp_iter>() : nullptr;
    sk.gn_trace = nullptr;
    sk.cov_hx = p->cov_findif_xyz;
    sk.cov_ha = p->cov_findif_ang;
    // Streaming loop control (single alignments with automatic polling): instead of a predicted chunk of iterations whose
    // unused tail idles on the stream (27.8 iterations' worth of kernels enqueued for 21 executed on the city drive, plus
    // 0.64 extra host round trips per scan), the host follows the progress word the solve kernels publish and keeps
    // MH_STREAM_LEAD (2) iterations queued ahead.  Not for lock-step batches (defer_upload), profiled jobs, or when switched
    // off (MH_NO_STREAM=1).
    streaming = p->poll_every == 0 && !defer_upload && !prof && ctx->d_progress != nullptr && getenv("MH_NO_STREAM") == nullptr;
    sk.host_progress = streaming ? ctx->d_progress : nullptr;
    nb = nblk(scan->n);
    {  // MH_MATCH selects the correspondence kernel of the fused loop (all exact, bit-identical pairings):
       //   "q"            a DPP quad per scan point, merged candidate scans (the default of rounds 1-4) -> k_match4 + k_accum
       //   "p"            one lane per point, branch-and-bound, fused accumulation   -> k_match<true, 1>
       //   "x"            one lane per point, the literal 27-voxel scan of the reference (A/B baseline)
      //   "s"            a DPP row (16 lanes) per point: what "q" becomes automatically for small layers    -> k_match16 + k_accum
      //   "t"            a workgroup per tile of the spatially sorted scan, map records staged in LDS      -> k_match_tile + k_accum
      const char* e = getenv("MH_MATCH");
      variant = scan->n <= kRowMaxPoints ? 5 : 9;
      //   "w"            a wave per tile of 64 sorted points, wave-uniform candidates through the scalar path       -> k_match_wave + k_accum
      //   "o"            "q" over the scan in search order (the sort of "t"/"w", no tiles)                                    -> k_match4 + k_accum
      //   ("t", "w", "o": mh_dev_variants.h, the development library only)
#ifdef MH_DEV_VARIANTS
      if (e && e[0] == 't') variant = 6;
      if (e && e[0] == 'w') variant = 7;
      if (e && e[0] == 'o') variant = 8;
#else
      if (e && (e[0] == 't' || e[0] == 'w' || e[0] == 'o'))
        return fail(MH_ERR_INVALID_ARGUMENT, "MH_MATCH=%c names a development matcher: build tools/variants/libmolahip_dev.so (tools/build_variants.sh)", e[0]);
#endif
      if (e && e[0] == 'q') variant = 4;
      //   "f" (default)  plan / scan (mh_nn_flat.h): a wave per 64 points, per-point / per-voxel / per-record work each on all 64 lanes -> k_match_flat + k_accum
      if (e && e[0] == 'f') variant = 9;
      if (e && e[0] == 's') variant = 5;
      if (e && e[0] == 'p') variant = 0;
      if (e && e[0] == 'x') variant = 1;
      if (variant == 1 && map->view().ndt) variant = 0;  // "x" walks contiguous z-runs; NDT maps interleave statistics records
      // MH_MATCHED_POINTS_SKIP (U12): the point matcher has to know the plane matcher's verdict for the same point -- the row
      // kernel runs both in one launch (k_match16<true>), whatever the layer's size
      if (pl && p->matched_points == MH_MATCHED_POINTS_SKIP) variant = 5;
    }
    if (variant == 9 && map->pts.bytes / sizeof(float4) >= kFlatMaxRecords) variant = 4;  // (the chunk word holds 30 bits of record index)
#ifdef MH_DEV_VARIANTS
    if (variant >= 6 && variant != 9) MH_TRY(scan_build_tiles(scan, map->inv_vs, variant == 7 ? 64u : 256u));
#endif
    if (variant == 4 || variant >= 6) {  // nn_search_quad / the plan-scan matcher read the map's sub-voxel index
      MH_TRY(map_ensure_qidx(map, ctx->stream));
      if (!map->view().pts_q) return fail(MH_ERR_INTERNAL, "the map's sub-voxel index is missing (matcher variant %d needs it)", variant);
    }
    nba = nblk_acc(scan->n);
    // (measured per iteration, fused vs k_accum: 31.0 vs 33.8 us at 4 k points, 33.2 vs 35.0 at 8 k, equal at 16 k,
    //  45.9 vs 41.8 at 32 k -- one partial row per 16 points makes the solve's reduction the longer pole there)
    fused16 = variant == 5 && !pl && scan->n <= kFused16MaxPoints && getenv("MH_NO_FUSE16") == nullptr;
    // who writes the partials of the first Gauss-Newton step: the row kernel (16 points per workgroup), k_accum, or k_match
    nbm = fused16 ? (uint32_t)((16ull * scan->n + kBlock - 1) / kBlock) : (variant >= 4 ? nba : nb);
    MH_TRY(ctx->partials.reserve((size_t)kGenN * (nbm > nb ? nbm : nb) * sizeof(double)));
    if (variant == 5 && scan->n <= kStepMaxPoints) {  // k_step16: two halves of one column per group of 32 points
      const size_t ng = (scan->n + kStepPoints - 1) / kStepPoints;
      MH_TRY(ctx->partials.reserve(2 * (size_t)kStepRowsA * (ng ? ng : 1) * sizeof(double)));
      if (pl) MH_TRY(ctx->partials_b.reserve(2 * (size_t)kStepRowsB * (ng ? ng : 1) * sizeof(double)));
    }
    step_src = 2;
    step_ppar = 0;
    // k_icp16: where the streaming chain would run (a single alignment under automatic control) and the layer has at most
    // kLoopMaxGroups groups -- if the workgroups of all loops running on this device still fit its CUs (every workgroup of a
    // loop has to be resident while it runs); otherwise the chain, bit for bit the same result
    loop16 = false;
    // (after a loop had to be abandoned -- somebody else's work kept its workgroups from running together, ~0.1 s lost -- the
    //  next 200 alignments on the device take the chain before another loop is tried)
    if (streaming && use_step_chain() && !forbid_loop16 && getenv("MH_NO_LOOP16") == nullptr && !loop_holdoff(ctx->device, false)) {
      const uint32_t ng = (uint32_t)((scan->n + kStepPoints - 1) / kStepPoints);
      loopw = !pl && loop_wave_enabled();
      if (ng <= (loopw ? kLwMaxGroups : kLoopMaxGroups) && loop_admit(ctx->device, loop_units(ng, pl))) {
        loop_wgs = loop_units(ng, pl);
        loop16 = true;
        streaming = false;
        sk.host_progress = nullptr;
        if (loopw) {  // the plan / scan search reads the map's sub-voxel index (built once per map content, on this stream)
          mh_status qs = map_ensure_qidx(map, ctx->stream);
          if (qs == MH_OK && !map->view().pts_q) qs = fail(MH_ERR_INTERNAL, "the map's sub-voxel index is missing (k_icpw needs it)");
          if (qs != MH_OK) {
            loop_release();
            return qs;
          }
        }
        if (ctx->loop_x.bytes < kLoopExchangeBytes) {
          const mh_status rs = ctx->loop_x.reserve(kLoopExchangeBytes);
          if (rs != MH_OK) {
            loop_release();
            return rs;
          }
          (void)hipMemsetAsync(ctx->loop_x.p, 0, kLoopExchangeBytes, s);
        }
      }
    }
    ws = nullptr;
    if (loop16 && !defer_upload && getenv("MH_NO_LOOP_STREAM") == nullptr) {
      const mh_status ls = loop_stream_behind(ctx, &ws);
      if (ls != MH_OK) {
        loop_release();
        return ls;
      }
    }
    if (defer_upload) {
      ctx->h_params->mk = mk;
      ctx->h_params->sk = sk;
    } else {
      MH_TRY(upload_state_and_params(ctx, mk, sk, inline_sched, ws));
    }
    // poll_every == 0: the first chunk is sized by what the previous alignment of this context needed (consecutive scans
    // of a sequence converge in about as many iterations: one host round trip instead of three), later chunks are short
    auto_chunk = p->poll_every == 0;
    // Every early-exit launch enqueued beyond the end of the loop costs ~1.5 us of stream time and every extra host poll
    // ~25 us, so the first chunk should be as long as the loop will run: the caller's estimate when it has one (the
    // odometry driver's calls alternate between short ones that the hook stops and long ones that converge, and it knows
    // which kind it is making), else what this context's previous alignment needed.
    kind = 0;
    {
      const uint32_t expect = p->expected_iterations ? p->expected_iterations : ctx->predicted_iterations[kind];
      static const uint32_t margin = getenv("MH_CHUNK_MARGIN") ? (uint32_t)atoi(getenv("MH_CHUNK_MARGIN")) : 2u;
      chunk = p->poll_every ? p->poll_every : (expect ? (expect + margin > 64 ? 64u : expect + margin) : 10u);
    }
    polls = 0;
    enqueued = 0;
    prof_n = 0;
    if (prof) {
      const uint32_t need = 2 * p->max_iterations;
      if (ctx->prof_cap < need) {
        hipEvent_t* ne = new (std::nothrow) hipEvent_t[need];
        if (!ne) return fail(MH_ERR_OUT_OF_MEMORY, "host allocation failed");
        for (uint32_t i = 0; i < ctx->prof_cap; i++) ne[i] = ctx->prof_ev[i];
        for (uint32_t i = ctx->prof_cap; i < need; i++) MH_HIP(hipEventCreate(&ne[i]));
        delete[] ctx->prof_ev;
        ctx->prof_ev = ne;
        ctx->prof_cap = need;
      }
      MH_HIP(hipEventRecord(ctx->ev_t0, s));
    }
    return MH_OK;
  }

  // issues the copies a deferred start() left out, on this job's own stream
  mh_status flush_deferred() {
    if (!defer_upload || finished) return MH_OK;
    MH_TRY(set_device(ctx));
    MH_HIP(hipMemcpyAsync(ctx->sched.p, ctx->h_sched, nsched_pending * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    MH_HIP(hipMemcpyAsync(ctx->d_state, ctx->h_state, kParamsOffset + sizeof(IcpDeviceParams), hipMemcpyHostToDevice,
                          ctx->stream));
    defer_upload = false;
    return MH_OK;
  }

  // Admission of one-launch loops: the workgroups of all loops running on a device must be resident together (a loop's
  // workgroups wait for each other), so their sum stays below the CU count less a reserve for what else is running; kernels
  // that merely pass through (other sequences' large layers, map updates) delay a loop's start, they cannot block it.
  static std::atomic<uint32_t>& loop_count(int device) {
    static std::atomic<uint32_t> c[64];
    return c[(unsigned)device % 64u];
  }
  static bool loop_admit(int device, uint32_t wgs) {  // wgs: in half CUs (loop_units)
    const uint32_t limit = loop_cu_limit(device) * kLoopUnitsPerCu;
    if (wgs > limit) return false;
    // A loop that does not fit now waits for the running ones (each takes a fraction of a millisecond) rather than fall back to
    // the chain, whose forty-odd launches would queue behind the same loops: eight sequences of the default pipeline keep five
    // loops on the device at any time.  MH_LOOP16_WAIT_US (2000): how long before the chain is taken after all.
    static const long wait_us = getenv("MH_LOOP16_WAIT_US") ? atol(getenv("MH_LOOP16_WAIT_US")) : 2000L;
    std::atomic<uint32_t>& c = loop_count(device);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; spins++) {
      uint32_t cur = c.load();
      while (cur + wgs <= limit)
        if (c.compare_exchange_weak(cur, cur + wgs)) return true;
      if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= wait_us) return false;
      if (spins < 64) __builtin_ia32_pause();
      else std::this_thread::yield();
    }
  }
  static bool loop_holdoff(int device, bool arm) {  // arm: a loop was abandoned; else: true while the hold-off lasts (counts down)
    static std::atomic<int> left[64];
    std::atomic<int>& v = left[(unsigned)device % 64u];
    if (arm) {
      v.store(200);
      return true;
    }
    int cur = v.load();
    while (cur > 0)
      if (v.compare_exchange_weak(cur, cur - 1)) return true;
    return false;
  }
  void loop_release() {
    if (loop_wgs) loop_count(ctx->device).fetch_sub(loop_wgs);
    loop_wgs = 0;
  }
  ~AlignJob() { loop_release(); }

  // The whole loop in one launch, then the covariance kernels + the state read-back, one event wait.  Leaves `finished`
  // false when the loop did not run to its end (a workgroup gave up waiting for the others): the caller starts over with the
  // launch-by-launch chain.
  mh_status run_loop16() {
    MH_TRY(set_device(ctx));
    hipStream_t s = ws ? ws : ctx->stream;
    const uint32_t n = (uint32_t)scan->n;
    const uint32_t ngr = (n + kStepPoints - 1) / kStepPoints;
    // (MH_LOOP16_TEST_ABANDON: the loop is cut short as if its workgroups had given up -- the caller's second attempt is what is tested)
    const uint32_t max_steps = getenv("MH_LOOP16_TEST_ABANDON") ? 1u : p->max_iterations * p->gn.max_inner_iterations + 1u;
    const uint32_t serial0 = ctx->loop_serial;
    ctx->loop_serial += max_steps + 2u;
    g_loop16_runs.fetch_add(1);
    char* const xa = static_cast<char*>(ctx->loop_x.p);
    char* const xb = xa + 2 * (size_t)kAccN * kLoopRowStride * 16;
    const MapView mv = map->view();
    if (loopw)
      hipLaunchKernelGGL(k_icpw, dim3((ngr + kLwGroups - 1) / kLwGroups), dim3(kLwThreads), 0, s, ctx->d_state, &ctx->d_params->mk,
                         &ctx->d_params->sk, scan->x, scan->y, scan->z, n, mv, ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(),
                         (void*)xa, ngr, serial0, max_steps, p->compute_covariance ? 1u : 0u);
    else if (pl)
      hipLaunchKernelGGL(k_icp16<true>, dim3(ngr), dim3(kSolveThreads), 0, s, ctx->d_state, &ctx->d_params->mk, &ctx->d_params->sk, scan->x,
                         scan->y, scan->z, n, mv, ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(), ctx->pl_c.as<float4>(),
                         ctx->pl_n.as<float4>(), (void*)xa, (void*)xb, ngr, serial0, max_steps, p->compute_covariance ? 1u : 0u);
    else
      hipLaunchKernelGGL(k_icp16<false>, dim3(ngr), dim3(kSolveThreads), 0, s, ctx->d_state, &ctx->d_params->mk, &ctx->d_params->sk, scan->x,
                         scan->y, scan->z, n, mv, ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(), (float4*)nullptr,
                         (float4*)nullptr, (void*)xa, (void*)xb, ngr, serial0, max_steps, p->compute_covariance ? 1u : 0u);
    enqueued = p->max_iterations;
    skip_tail = false;
    MH_TRY(enqueue_tail(/*cov_prepared=*/true, s));
    const hipError_t we = mh::wait_event(ctx->ev_poll);
    loop_release();
    ws = nullptr;  // (whatever follows -- the chain after an abandoned loop, the pairing export -- runs on the context's own stream: the host has waited)
    MH_HIP(we);
    const IcpDeviceState* h = ctx->h_state;
    if (!h->done || h->handover_timeouts) {
      g_loop16_fallbacks.fetch_add(1);
      if (getenv("MH_LOOP16_TEST_ABANDON") == nullptr) loop_holdoff(ctx->device, true);
      return MH_OK;  // (not finished)
    }
    return poll(true);
  }

  // row-kernel layers up to kStepMaxPoints: k_step16 launches (profiled jobs time the match kernel alone -- they, and
  // MH_NO_STEP_CHAIN=1, take the chains with a match kernel of its own)
  bool use_step_chain() const {
    return variant == 5 && scan->n <= kStepMaxPoints && !prof && getenv("MH_NO_STEP_CHAIN") == nullptr;
  }

  mh_status enqueue_chunk() {
    if (finished) return MH_OK;
    MH_TRY(set_device(ctx));
    hipStream_t s = ctx->stream;
    const uint32_t n = (uint32_t)scan->n;
    const MapView mv = map->view();
#ifdef MH_DEV_VARIANTS
    if (variant == 6 || variant == 7) MH_TRY(scan_tiles_ready(scan));  // the launch grid needs the tile count
#endif
    const uint32_t m = (p->max_iterations - enqueued) < chunk ? (p->max_iterations - enqueued) : chunk;
    PoseArg dummy{};
    double* part = ctx->partials.as<double>();
    const MatchK* dmk = &ctx->d_params->mk;
    const SolveK* dsk = &ctx->d_params->sk;
    // everything a chunk launches, in stream order; used directly (profiling / MH_NO_GRAPH) or under stream capture
    const bool step_chain = use_step_chain();  // k_step16: every launch over the whole layer, the solve carried into the next launch
    const uint32_t ngr = n ? (n + kStepPoints - 1) / kStepPoints : 1u;
    const uint32_t nwg = ngr < kStepMaxWorkgroups ? ngr : kStepMaxWorkgroups;
    // (launches whose arguments may be frozen in a captured graph cannot be told their number)
    const bool counted = prof || streaming || getenv("MH_NO_GRAPH") != nullptr;
    uint32_t chunk_k = 0;  // k_step16 launches of this chunk so far
    auto launch_step = [&](uint32_t close_only) {
      const uint32_t expect = counted ? serial_base + step_launches : chunk_k;  // (replayed: the place in the chunk)
      const uint32_t expect_rel = counted ? 0u : 1u;
      step_launches++;
      chunk_k++;
      // canonical block (upload, results) + a ping-pong pair: a launch never writes the block it reads, and the canonical one is only
      // written by launches that do not read it (the one that ends the loop, the one-workgroup close-only launch)
      IcpDeviceState* const S[3] = {ctx->d_state_b, reinterpret_cast<IcpDeviceState*>(reinterpret_cast<char*>(ctx->d_state_b) + 256), ctx->d_state};
      const uint32_t src = step_src, dst = close_only ? 2u : (src == 2u ? 0u : (src ^ 1u));
      double* const pa[2] = {part, part + (size_t)kStepRowsA * ngr};
      double* const pbb = pl ? ctx->partials_b.as<double>() : nullptr;
      double* const pb[2] = {pbb, pbb ? pbb + (size_t)kStepRowsB * ngr : nullptr};
      const uint32_t par = step_ppar;
      if (pl)
        hipLaunchKernelGGL(k_step16<true>, dim3(close_only ? 1u : nwg), dim3(kSolveThreads), 0, s, S[src], S[dst],
                           S[2], dmk, dsk, scan->x, scan->y, scan->z, n, mv, ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(),
                           ctx->pl_c.as<float4>(), ctx->pl_n.as<float4>(), (const double*)pa[par], pa[par ^ 1u], (const double*)pb[par],
                           pb[par ^ 1u], ngr, close_only, expect, expect_rel);
      else
        hipLaunchKernelGGL(k_step16<false>, dim3(close_only ? 1u : nwg), dim3(kSolveThreads), 0, s, S[src], S[dst],
                           S[2], dmk, dsk, scan->x, scan->y, scan->z, n, mv, ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(),
                           (float4*)nullptr, (float4*)nullptr, (const double*)pa[par], pa[par ^ 1u], (const double*)nullptr,
                           (double*)nullptr, ngr, close_only, expect, expect_rel);
      step_src = dst;
      if (!close_only) step_ppar = par ^ 1u;
    };
    auto enqueue_kernels = [&]() -> mh_status {
      double* partb = pl ? ctx->partials_b.as<double>() : nullptr;
      const bool rows16 = pl && variant == 5;                  // NDT layer handled by the row kernel
      const uint32_t nB = pl ? (rows16 ? nba : nb) : 0u;       // columns of the point-to-plane partials of the FIRST step
      const uint32_t nBi = pl ? nba : 0u;                      // ... of the inner steps (k_accum_both)
      for (uint32_t j = 0; j < m; j++) {
        if (step_chain) {
          for (uint32_t in = 0; in < p->gn.max_inner_iterations; in++) launch_step(0u);
          continue;
        }
        const bool both16 = pl && variant == 5;  // small layer: both matchers in one launch (k_match16<true>)
        if (pl && !both16)
          hipLaunchKernelGGL(k_match_pl<true>, dim3(nb), dim3(kBlock), 0, s, ctx->d_state, dummy, 0.f, dmk, scan->x, scan->y,
                             scan->z, n, mv, ctx->pl_c.as<float4>(), ctx->pl_n.as<float4>(), partb, nb);
        if (prof) MH_HIP(hipEventRecord(ctx->prof_ev[2 * prof_n], s));
        if (variant == 5) {
          if (fused16)
            hipLaunchKernelGGL((k_match16<false, true>), dim3(nbm), dim3(kBlock), 0, s, ctx->d_state, dmk, scan->x, scan->y,
                               scan->z, n, mv, ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(), (float4*)nullptr,
                               (float4*)nullptr, part, nbm);
          else if (both16)
            hipLaunchKernelGGL((k_match16<true, false>), dim3((uint32_t)((16ull * n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, ctx->d_state,
                               dmk, scan->x, scan->y, scan->z, n, mv, ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(),
                               ctx->pl_c.as<float4>(), ctx->pl_n.as<float4>(), (double*)nullptr, 0u);
          else
            hipLaunchKernelGGL((k_match16<false, false>), dim3((uint32_t)((16ull * n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, ctx->d_state,
                               dmk, scan->x, scan->y, scan->z, n, mv, ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(),
                               (float4*)nullptr, (float4*)nullptr, (double*)nullptr, 0u);
          if (prof) MH_HIP(hipEventRecord(ctx->prof_ev[2 * prof_n + 1], s));  // the match kernel alone
          if (both16)  // both kinds of Gauss-Newton rows of the pairings just written
            hipLaunchKernelGGL(k_accum_both, dim3(nba), dim3(kBlock), 0, s, ctx->d_state, 1u, dmk, scan->x, scan->y, scan->z, n,
                               ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(), ctx->pl_c.as<float4>(),
                               ctx->pl_n.as<float4>(), part, partb, nba);
          else if (!fused16)
            hipLaunchKernelGGL(k_accum<false>, dim3(nba), dim3(kBlock), 0, s, ctx->d_state, 1u, dmk, scan->x, scan->y, scan->z, n,
                               ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(), part, nbm);
#ifdef MH_DEV_VARIANTS
        } else if (variant == 7) {
#ifdef MH_DEBUG_WAVETRACE
          MH_LAUNCH_WAVE(s, ctx->d_state, scan, mv, ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(), MH_WT_G);
#else
          MH_LAUNCH_WAVE(s, ctx->d_state, scan, mv, ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(), );
#endif
          if (prof) MH_HIP(hipEventRecord(ctx->prof_ev[2 * prof_n + 1], s));  // the match kernel alone
          hipLaunchKernelGGL(k_accum<false>, dim3(nba), dim3(kBlock), 0, s, ctx->d_state, 1u, dmk, scan->x, scan->y, scan->z, n,
                             ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(), part, nba);
        } else if (variant == 6) {
          hipLaunchKernelGGL(k_match_tile, dim3(scan->n_tiles), dim3(kTileThreads), 0, s, ctx->d_state, scan->sx, scan->sy, scan->sz,
                             scan->perm, scan->tile_start, scan->n_tiles, mv, ctx->pair_q.as<float4>(),
                             ctx->pair_gidx.as<uint32_t>()
#ifdef MH_DEBUG_WAVETRACE
                             , g_wtrace
#endif
          );
          if (prof) MH_HIP(hipEventRecord(ctx->prof_ev[2 * prof_n + 1], s));  // the match kernel alone
          hipLaunchKernelGGL(k_accum<false>, dim3(nba), dim3(kBlock), 0, s, ctx->d_state, 1u, dmk, scan->x, scan->y, scan->z, n,
                             ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(), part, nba);
#endif
        } else if (variant == 9) {
          hipLaunchKernelGGL(k_match_flat, dim3(nblk_flat(n)), dim3(kFlatThreads), 0, s, ctx->d_state, scan->x, scan->y, scan->z, n, mv,
                             ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(), (const uint32_t*)nullptr);
          if (prof) MH_HIP(hipEventRecord(ctx->prof_ev[2 * prof_n + 1], s));  // the match kernel alone
          hipLaunchKernelGGL(k_accum<true>, dim3(nba), dim3(kBlock), 0, s, ctx->d_state, 1u, dmk, scan->x, scan->y, scan->z, n,
                             ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(), part, nba);
        } else if (variant == 4 || variant == 8) {
#ifdef MH_DEV_VARIANTS
          const bool ord = variant == 8;  // the scan in search order
#else
          const bool ord = false;
#endif
          hipLaunchKernelGGL(k_match4, dim3((uint32_t)((4ull * n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, ctx->d_state,
                             ord ? scan->sx : scan->x, ord ? scan->sy : scan->y, ord ? scan->sz : scan->z, n, mv,
                             ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(), ord ? scan->perm : (const uint32_t*)nullptr
#ifdef MH_DEBUG_WAVETRACE
                             , g_wtrace
#endif
          );
          if (prof) MH_HIP(hipEventRecord(ctx->prof_ev[2 * prof_n + 1], s));  // the match kernel alone
          hipLaunchKernelGGL(k_accum<false>, dim3(nba), dim3(kBlock), 0, s, ctx->d_state, 1u, dmk, scan->x, scan->y, scan->z, n,
                             ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(), part, nba);
        } else if (variant == 1)
          hipLaunchKernelGGL((k_match<true, 0>), dim3(nb), dim3(kBlock), 0, s, ctx->d_state, dummy, 0.f, 1u, dmk, scan->x,
                             scan->y, scan->z, n, mv, ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(), part, nbm);
        else
          hipLaunchKernelGGL((k_match<true, 1>), dim3(nb), dim3(kBlock), 0, s, ctx->d_state, dummy, 0.f, 1u, dmk, scan->x,
                             scan->y, scan->z, n, mv, ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(), part, nbm);
        if (prof) {
          if (variant < 4) MH_HIP(hipEventRecord(ctx->prof_ev[2 * prof_n + 1], s));
          prof_n++;
        }
        hipLaunchKernelGGL(k_solve, dim3(1), dim3(kSolveThreads), 0, s, ctx->d_state, dsk, part, nbm, nbm,
                           (const double*)partb, nB, nB, 1u);
        for (uint32_t in = 1; in < p->gn.max_inner_iterations; in++) {
          if (pl)
            hipLaunchKernelGGL(k_accum_both, dim3(nba), dim3(kBlock), 0, s, ctx->d_state, 0u, dmk, scan->x, scan->y, scan->z, n,
                               ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(), ctx->pl_c.as<float4>(),
                               ctx->pl_n.as<float4>(), part, partb, nba);
          else
            hipLaunchKernelGGL(variant == 9 ? k_accum<true> : k_accum<false>, dim3(nba), dim3(kBlock), 0, s, ctx->d_state, 0u, dmk, scan->x, scan->y, scan->z, n,
                               ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(), part, nba);
          hipLaunchKernelGGL(k_solve, dim3(1), dim3(kSolveThreads), 0, s, ctx->d_state, dsk, part, nba, nba,
                             (const double*)partb, nBi, nBi, 0u);
        }
      }
      // the step chain's last Gauss-Newton step is still pending: at the end of a chunk the host is going to look at the state,
      // under streaming control only once the iteration budget is queued (a loop that ends earlier is seen by the next launch)
      if (step_chain && (!skip_tail || enqueued + m >= p->max_iterations)) launch_step(1u);
      if (skip_tail) return MH_OK;  // streaming: the tail below is enqueued once, by enqueue_tail()
      if (p->compute_covariance) {  // no-ops unless the loop has terminated
        hipLaunchKernelGGL(k_cov_prepare, dim3(1), dim3(64), 0, s, ctx->d_state, dsk, 0u);
        hipLaunchKernelGGL(k_cov_accum, dim3(nb), dim3(kBlock), 0, s, ctx->d_state, 0u, scan->x, scan->y, scan->z, n,
                           ctx->pair_gidx.as<uint32_t>(), part, nb);
        if (pl)
          hipLaunchKernelGGL(k_cov_accum_plbuf, dim3(nb), dim3(kBlock), 0, s, ctx->d_state, scan->x, scan->y, scan->z, n,
                             ctx->pl_c.as<float4>(), ctx->pl_n.as<float4>(), partb, nb);
        hipLaunchKernelGGL(k_cov_finalize, dim3(1), dim3(kSolveThreads), 0, s, ctx->d_state, 0u, part, nb, nb,
                           (const double*)partb, pl ? nb : 0u, pl ? nb : 0u);  // (the covariance kernels write nb columns)
      }
      MH_HIP(hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(IcpDeviceState), hipMemcpyDeviceToHost, s));
      return MH_OK;
    };
    const bool no_graph = getenv("MH_NO_GRAPH") != nullptr;
    if (prof || no_graph || streaming) {
      MH_TRY(enqueue_kernels());
      MH_HIP(hipGetLastError());
      if (streaming) {  // (no event per iteration: the progress word is the signal)
        enqueued += m;
        return MH_OK;
      }
    } else {
      if (step_chain) {  // the serial number this chunk's first k_step16 launch has to find (its launches carry their place in the chunk)
        ctx->h_params->sk.step_base = serial_base + step_total;
        MH_HIP(hipMemcpyAsync(&ctx->d_params->sk.step_base, &ctx->h_params->sk.step_base, sizeof(uint32_t), hipMemcpyHostToDevice, s));
        step_total += m * p->gn.max_inner_iterations + ((!skip_tail || enqueued + m >= p->max_iterations) ? 1u : 0u);
      }
      // The launch sequence only depends on sizes and device pointers (the per-alignment values sit in device
      // memory), so it is captured once and replayed: one host call per chunk instead of ~4 per iteration.
      unsigned long long key[32] = {0};
      uint32_t fb;
      memcpy(&fb, &mv.inv_vs, 4);
      const unsigned long long kv[] = {n, nb, nbm, (unsigned long long)variant, m, p->gn.max_inner_iterations,
                                       p->compute_covariance, (unsigned long long)mv.slots, (unsigned long long)mv.pts, mv.mask,
                                       fb, mv.trunc | (mv.ndt << 1) | (mv.no_prev_bound << 2), (unsigned long long)scan->x, (unsigned long long)scan->y,
                                       (unsigned long long)scan->z, (unsigned long long)ctx->pair_q.p,
                                       (unsigned long long)ctx->pair_gidx.p, (unsigned long long)part,
                                       (unsigned long long)ctx->d_state, (unsigned long long)ctx->d_params,
                                       (unsigned long long)ctx->h_state,
                                       (pl ? 2ull : 1ull) | (fused16 ? 8ull : 0ull) | (step_chain ? 16ull : 0ull),
                                       (unsigned long long)(pl ? ctx->pl_c.p : nullptr),
                                       (unsigned long long)(pl ? ctx->pl_n.p : nullptr),
                                       (unsigned long long)(pl ? ctx->partials_b.p : nullptr),
                                       variant >= 6 && variant != 9 ? (unsigned long long)scan->sx : (unsigned long long)variant,
                                       variant >= 6 && variant != 9 ? (unsigned long long)scan->n_tiles : 0ull,
                                       (unsigned long long)mv.pts_q};  // (a word each: no XOR folding)
      static_assert(sizeof(kv) <= sizeof(key), "graph key too small");
      memcpy(key, kv, sizeof(kv));
      const bool cached = ctx->graph_exec && memcmp(key, ctx->graph_key, sizeof(key)) == 0;
      const bool seen_before = memcmp(key, ctx->graph_candidate, sizeof(key)) == 0 && ctx->graph_candidate_align != ctx->align_serial;
      if (!cached && !seen_before) {
        // a shape not seen in an earlier alignment: launch directly and remember it; it is captured when a later
        // alignment brings it again.  (The real pipeline's ICP layer changes size with every scan: capturing and
        // instantiating a graph per alignment cost 0.4 ms each.)
        if (memcmp(key, ctx->graph_candidate, sizeof(key)) != 0) {
          memcpy(ctx->graph_candidate, key, sizeof(key));
          ctx->graph_candidate_align = ctx->align_serial;
        }
        MH_TRY(enqueue_kernels());
        MH_HIP(hipGetLastError());
        enqueued += m;
        MH_HIP(hipEventRecord(ctx->ev_poll, s));
        return MH_OK;
      }
      if (!cached) {
        if (ctx->graph_exec) {
          (void)hipGraphExecDestroy(ctx->graph_exec);
          ctx->graph_exec = nullptr;
        }
        hipGraph_t g = nullptr;
        MH_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        const mh_status cs = enqueue_kernels();
        const hipError_t ce = hipStreamEndCapture(s, &g);
        if (cs != MH_OK) {
          if (g) (void)hipGraphDestroy(g);
          return cs;
        }
        if (ce != hipSuccess) return fail(MH_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(ce));
        const hipError_t ie = hipGraphInstantiate(&ctx->graph_exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ie != hipSuccess) {
          ctx->graph_exec = nullptr;
          return fail(MH_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(ie));
        }
        memcpy(ctx->graph_key, key, sizeof(key));
      }
      MH_HIP(hipGraphLaunch(ctx->graph_exec, s));
    }
    enqueued += m;
    if (prof) MH_HIP(hipEventRecord(ctx->ev_t1, s));
    MH_HIP(hipEventRecord(ctx->ev_poll, s));
    return MH_OK;
  }

  // The whole loop of a single alignment under streaming control: iterations are enqueued one at a time, at most `lead`
  // ahead of the iteration the device has published; once it publishes "done" the covariance kernels and the state
  // read-back follow and ONE event wait ends the call.
  mh_status run_streaming() {
    static const uint32_t lead0 = getenv("MH_STREAM_LEAD") ? (uint32_t)std::max(1, atoi(getenv("MH_STREAM_LEAD"))) : 2u;
    const uint32_t lead = lead0 + (use_step_chain() ? 1u : 0u);  // (k_step16 publishes an iteration's end from the NEXT launch)
    volatile uint32_t* prog = ctx->h_progress;
    *prog = 0;  // (the previous alignment of this context has been waited for: nothing in flight writes it)
    chunk = 1;
    skip_tail = true;
    uint32_t spins = 0;
    for (;;) {
      const uint32_t pv = *prog;
      if (pv >> 31) break;
      const uint32_t dev_iter = pv & 0x7FFFFFFFu;
      if (enqueued < p->max_iterations && enqueued < dev_iter + lead) {
        MH_TRY(enqueue_chunk());
        spins = 0;
        continue;
      }
      __builtin_ia32_pause();
      if (++spins > (1u << 22)) {  // ~ tens of milliseconds without progress: is the stream still alive?
        spins = 0;
        const hipError_t q = hipStreamQuery(ctx->stream);
        if (q != hipSuccess && q != hipErrorNotReady) return fail(MH_ERR_HIP, "device ICP loop: %s", hipGetErrorString(q));
        if (q == hipSuccess && !(*prog >> 31) && enqueued >= p->max_iterations)
          return fail(MH_ERR_INTERNAL, "device ICP loop did not terminate after max_iterations");
      }
    }
    skip_tail = false;
    chunk = 0;
    MH_TRY(enqueue_tail());  // the tail alone: covariance (now live: the loop has ended) + state read-back
    return poll();
  }

  mh_status enqueue_tail(bool cov_prepared = false, hipStream_t on = nullptr) {  // cov_prepared: k_icp16 has done k_cov_prepare's part
    MH_TRY(set_device(ctx));
    hipStream_t s = on ? on : ctx->stream;
    const uint32_t n = (uint32_t)scan->n;
    double* part = ctx->partials.as<double>();
    double* partb = pl ? ctx->partials_b.as<double>() : nullptr;
    const SolveK* dsk = &ctx->d_params->sk;
    if (p->compute_covariance) {
      if (!cov_prepared) hipLaunchKernelGGL(k_cov_prepare, dim3(1), dim3(64), 0, s, ctx->d_state, dsk, 0u);
      hipLaunchKernelGGL(k_cov_accum, dim3(nb), dim3(kBlock), 0, s, ctx->d_state, 0u, scan->x, scan->y, scan->z, n,
                         ctx->pair_gidx.as<uint32_t>(), part, nb);
      if (pl)
        hipLaunchKernelGGL(k_cov_accum_plbuf, dim3(nb), dim3(kBlock), 0, s, ctx->d_state, scan->x, scan->y, scan->z, n,
                           ctx->pl_c.as<float4>(), ctx->pl_n.as<float4>(), partb, nb);
      hipLaunchKernelGGL(k_cov_finalize, dim3(1), dim3(kSolveThreads), 0, s, ctx->d_state, 0u, part, nb, nb,
                         (const double*)partb, pl ? nb : 0u, pl ? nb : 0u);
    }
    MH_HIP(hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(IcpDeviceState), hipMemcpyDeviceToHost, s));
    MH_HIP(hipGetLastError());
    MH_HIP(hipEventRecord(ctx->ev_poll, s));
    return MH_OK;
  }

  // waits for the last enqueued chunk; sets finished when the device loop has terminated
  mh_status poll(bool already_synced = false) {  // (lock-step batches copy the state into h_state themselves)
    if (finished) return MH_OK;
    polls++;
    MH_TRY(set_device(ctx));
    if (!already_synced) MH_HIP(mh::wait_event(ctx->ev_poll));
    const IcpDeviceState* h = ctx->h_state;
    if (!h->done && enqueued < p->max_iterations) {
      if (auto_chunk) {
        static const uint32_t next = getenv("MH_CHUNK_NEXT") ? (uint32_t)atoi(getenv("MH_CHUNK_NEXT")) : 6u;
        chunk = next ? next : 6u;
      }
      return MH_OK;
    }
    if (h->handover_timeouts)
      return fail(MH_ERR_INTERNAL, "device ICP loop (k_step16): %u workgroup(s) gave up waiting for the state block or the partial sums of the previous launch [first: kind %u wg %u tid %u wanted %u (base %u) seen %u groups %u expect %u state(pending|iter|inner|done) %08x iter %u n %u streaming %d]",
                  h->handover_timeouts, h->dbg[0], h->dbg[1], h->dbg[2], h->dbg[3], serial_base, h->dbg[4], h->dbg[5], h->dbg[6], h->dbg[7], h->n_iterations, (uint32_t)scan->n, (int)streaming);
    if (!h->done) return fail(MH_ERR_INTERNAL, "device ICP loop did not terminate after max_iterations");
    finished = true;
    if (auto_chunk) ctx->predicted_iterations[kind] = h->n_iterations + (h->term_reason == MH_TERM_MAX_ITERATIONS ? 0u : 1u);
    res->n_host_polls = polls;
    res->n_enqueued_iterations = enqueued;
    for (int i = 0; i < 12; i++) res->T[i] = h->T[i];
    if (p->compute_covariance)
      for (int i = 0; i < 36; i++) res->cov[i] = h->cov[i];
    res->n_iterations = h->n_iterations;
    res->termination_reason = h->term_reason;
    res->n_final_pairs = h->n_pairs;
    res->n_final_pairs_pt2pl = pl ? h->n_pairs_pl : 0u;
    res->potential_pairings = scan->n * (pl ? 2u : 1u);
    res->quality = (h->n_pairs && scan->n) ? (double)h->n_pairs / (double)res->potential_pairings : 0.0;  // PairedRatio
    if (h->term_reason == MH_TERM_NO_PAIRINGS)
      for (int i = 0; i < 36; i++) res->cov[i] = (i % 7 == 0) ? 1e6 : 0.0;
    if (trace) {
      const uint32_t cnt = h->n_iterations < p->max_iterations ? h->n_iterations + 1 : p->max_iterations;
      memset(trace, 0, sizeof(mh_icp_iter) * p->max_iterations);
      const uint32_t valid = (h->term_reason == MH_TERM_NO_PAIRINGS || h->term_reason == MH_TERM_SOLVER_ERROR)
                                 ? h->n_iterations : cnt;
      if (valid) MH_HIP(hipMemcpy(trace, ctx->trace.p, sizeof(mh_icp_iter) * valid, hipMemcpyDeviceToHost));
    }
    if (prof) {
      float ms = 0.f;
      double sum = 0.0;
      // launches enqueued after termination are early-exit no-ops; time only the live ones
      uint32_t live = h->n_iterations + ((h->term_reason == MH_TERM_MAX_ITERATIONS) ? 0u : 1u);
      if (live > prof_n) live = prof_n;
      for (uint32_t i = 0; i < live; i++) {
        MH_HIP(hipEventElapsedTime(&ms, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]));
        sum += ms;
      }
      res->n_match_launches = live;
      res->match_kernel_ms = sum;
      MH_HIP(hipEventElapsedTime(&ms, ctx->ev_t0, ctx->ev_t1));
      res->total_ms = ms;
    }
    return MH_OK;
  }
};

mh_status check_align_args(const mh_map* map, const mh_scan* scan, const mh_icp_params* p, const double* T,
                           const mh_icp_result* res) {
  MH_REQUIRE(map && scan && p && T && res, "null argument");
  MH_REQUIRE(map->ctx->device == scan->ctx->device, "map and scan live on different devices");
  MH_REQUIRE(pose_ok(T), "non-finite initial guess");
  MH_REQUIRE(p->max_iterations == 0 || (p->threshold && p->kernel_param), "threshold/kernel_param arrays are required");
  MH_REQUIRE(p->gn.max_inner_iterations >= 1, "gn.max_inner_iterations must be >= 1");
  MH_REQUIRE(p->gn.robust_kernel <= MH_KERNEL_GM_C2, "unknown robust kernel");
  MH_REQUIRE(p->max_iterations < (1u << 20), "max_iterations too large");
  MH_REQUIRE(p->matched_points <= MH_MATCHED_POINTS_SKIP, "unknown matched_points mode");
  return MH_OK;
}

}  // namespace

extern "C" {

mh_status mh_icp_align(const mh_map* map, const mh_scan* scan, const mh_icp_params* params, const double T_guess[12],
                       const mh_prior* prior, mh_icp_result* result, mh_icp_iter* trace, const mh_pairs_out* final_pairs,
                       int32_t pairs_mem) {
  MH_TRY(check_align_args(map, scan, params, T_guess, result));
  MH_REQUIRE(!final_pairs || pairs_mem == MH_MEM_HOST || pairs_mem == MH_MEM_DEVICE, "bad mem space");
  AlignJob job;
  MH_TRY(job.start(map, scan, params, T_guess, prior, result, trace));
  if (job.loop16 && !job.finished) {
    MH_TRY(job.run_loop16());
    if (!job.finished) {  // the loop's workgroups did not all get to run together: once more, launch by launch
      job.forbid_loop16 = true;
      MH_TRY(job.start(map, scan, params, T_guess, prior, result, trace));
    }
  }
  if (job.streaming && !job.finished) MH_TRY(job.run_streaming());
  while (!job.finished) {
    MH_TRY(job.enqueue_chunk());
    MH_TRY(job.poll());
  }
  if (final_pairs && !job.trivial && result->n_final_pairs > result->n_final_pairs_pt2pl) {
    uint64_t np = 0;
    MH_TRY(compact_pairs(scan->ctx, scan->n, final_pairs, pairs_mem, &np));
    if (np != result->n_final_pairs - result->n_final_pairs_pt2pl)
      return fail(MH_ERR_INTERNAL, "pair compaction count mismatch: %llu pairings in the buffers, %u in the last accumulation",
                  (unsigned long long)np, result->n_final_pairs - result->n_final_pairs_pt2pl);
  }
  return MH_OK;
}

mh_status mh_icp_align_prefers_solo(const mh_scan* scan, const mh_icp_params* p, uint32_t concurrent_callers, int32_t* yes) {
  MH_REQUIRE(scan && p && yes, "null argument");
  // what AlignJob::start decides for a single alignment under automatic control: the row matcher's layer sizes (MH_MATCH unset or
  // "s"; always for MH_MATCHED_POINTS_SKIP), the k_step16 chain's conditions, at most kLoopMaxGroups groups
  const size_t n = scan->n;
  const bool pl = p->pt2pl_threshold != nullptr;
  const char* e = getenv("MH_MATCH");
  bool row = n <= kRowMaxPoints;
  if (e && (e[0] == 't' || e[0] == 'w' || e[0] == 'o' || e[0] == 'q' || e[0] == 'f' || e[0] == 'p' || e[0] == 'x')) row = false;
  if (e && e[0] == 's') row = true;
  if (pl && p->matched_points == MH_MATCHED_POINTS_SKIP) row = true;
  const size_t loop_max = (size_t)((!pl && loop_wave_enabled()) ? kLwMaxGroups : kLoopMaxGroups) * kStepPoints;
  *yes = (n > 0 && p->max_iterations > 0 && row && n <= loop_max && p->poll_every == 0 && p->profile == 0 &&
          scan->ctx->d_progress != nullptr && getenv("MH_NO_STREAM") == nullptr && getenv("MH_NO_STEP_CHAIN") == nullptr &&
          getenv("MH_NO_LOOP16") == nullptr)
             ? 1
             : 0;
  // ... and the loops of all callers fit the device together: nobody waits for a turn.  (70 % of the CUs: the callers' filters,
  // de-skew and map updates run beside the loops -- five loops of 44 workgroups on 256 CUs measured slower than four, 3780 against
  // 4000-4450 scans/s.)
  if (*yes && !loops_fit(scan->ctx->device, loop_units((uint32_t)((n + kStepPoints - 1) / kStepPoints), pl), concurrent_callers)) *yes = 0;
  return MH_OK;
}

int32_t mh_debug_dev_variants(void) {
#ifdef MH_DEV_VARIANTS
  return 1;
#else
  return 0;
#endif
}

void mh_debug_loop_stats(uint64_t* loops_started, uint64_t* loops_abandoned) {
  if (loops_started) *loops_started = g_loop16_runs.load();
  if (loops_abandoned) *loops_abandoned = g_loop16_fallbacks.load();
}

size_t mh_pairs_block_bytes(size_t n_scan_points) {
  const size_t S = ((n_scan_points ? n_scan_points : 1) + 63) / 64 * 64;
  return 6 * S * sizeof(uint32_t);
}

namespace {
// descriptor of one job for the *_b kernels (device pointers only; the pairing-block fields are filled by the caller)
void fill_batch_desc(const AlignJob& j, BatchJob& d) {
  memset(&d, 0, sizeof(d));
  d.st = j.ctx->d_state;
  d.st_b = j.ctx->d_state_b;
  d.serial_base = j.serial_base;
  d.mk = &j.ctx->d_params->mk;
  d.sk = &j.ctx->d_params->sk;
  d.lx = j.scan->x; d.ly = j.scan->y; d.lz = j.scan->z;
  d.n = (uint32_t)j.scan->n;
  d.nb = j.nb;
  d.nba = j.nba;
  d.nbm = j.nbm;
  d.map = j.map->view();
  d.pair_q = j.ctx->pair_q.as<float4>();
  d.pair_gidx = j.ctx->pair_gidx.as<uint32_t>();
  d.part = j.ctx->partials.as<double>();
  if (j.pl) {
    d.partb = j.ctx->partials_b.as<double>();
    d.pl_c = j.ctx->pl_c.as<float4>();
    d.pl_n = j.ctx->pl_n.as<float4>();
  }
  d.sched_dst = j.ctx->sched.as<uint32_t>();
  d.sched_dwords = (uint32_t)(2 * j.nsched_pending);
  if (j.variant >= 6 && j.variant != 9) {
    d.sx = j.scan->sx; d.sy = j.scan->sy; d.sz = j.scan->sz;
    d.perm = j.scan->perm;
    d.tile_start = j.scan->tile_start;
    d.n_tiles = j.scan->n_tiles;
  }
}

// Work still queued on a job's own stream (asynchronous uploads, filters, de-skew, an earlier alignment) must be
// complete before `lead`'s stream reads that job's scan / state: an event per job, waited for by the leader's stream.
mh_status order_after_job_streams(mh_ctx* lead, const std::vector<AlignJob*>& jobs) {
  for (AlignJob* j : jobs) {
    MH_TRY(map_ready_on(j->map, lead->stream));  // a key-frame update of this job's map still running on its side stream
    if (j->ctx == lead || j->ctx->stream == lead->stream) continue;
    if (hipStreamQuery(j->ctx->stream) == hipSuccess) continue;  // nothing pending there
    MH_HIP(hipEventRecord(j->ctx->ev_ready, j->ctx->stream));
    MH_HIP(hipStreamWaitEvent(lead->stream, j->ctx->ev_ready, 0));
  }
  (void)hipGetLastError();  // hipStreamQuery's hipErrorNotReady is not an error
  return MH_OK;
}

// Where the batch's final pairings go: device-side layout of the pairs block (header of per-block counts / offsets for
// the compaction + the block itself unless the caller's block already is device memory).
struct PairsPlan {
  bool want = false;
  int32_t mem = MH_MEM_HOST;
  char* host_block = nullptr;    // caller's block (host kinds)
  char* dev_block = nullptr;     // where the kernels write
  uint32_t* dev_hdr = nullptr;
  size_t total_bytes = 0;
  std::vector<size_t> off;       // byte offset of job i in the block
  std::vector<size_t> hdr_off;   // entry offset of job i's counts in the header
};

mh_status plan_pairs(mh_ctx* lead, const std::vector<AlignJob>& jobs, void* pairs_block, int32_t pairs_mem, PairsPlan& pp) {
  pp.want = pairs_block != nullptr;
  if (!pp.want) return MH_OK;
  pp.mem = pairs_mem;
  pp.off.resize(jobs.size());
  pp.hdr_off.resize(jobs.size());
  size_t bytes = 0, hdr = 0;
  for (size_t i = 0; i < jobs.size(); i++) {
    pp.off[i] = bytes;
    pp.hdr_off[i] = hdr;
    bytes += mh_pairs_block_bytes(jobs[i].scan->n);
    hdr += 2 * (size_t)nblk(jobs[i].scan->n ? jobs[i].scan->n : 1);
  }
  pp.total_bytes = bytes;
  const size_t hdr_bytes = (hdr * 4 + 255) / 256 * 256;
  const size_t need = hdr_bytes + (pairs_mem == MH_MEM_DEVICE ? 0 : bytes);
  if (lead->pairs_stage.bytes < need && lead->pairs_copy_pending) {  // the previous download still reads the old buffer
    MH_HIP(mh::wait_stream(lead->copy_stream));
    lead->pairs_copy_pending = false;
  }
  MH_TRY(lead->pairs_stage.reserve(need));
  pp.dev_hdr = lead->pairs_stage.as<uint32_t>();
  pp.dev_block = pairs_mem == MH_MEM_DEVICE ? (char*)pairs_block : lead->pairs_stage.as<char>() + hdr_bytes;
  pp.host_block = pairs_mem == MH_MEM_DEVICE ? nullptr : (char*)pairs_block;
  if (pairs_mem == MH_MEM_HOST_PINNED && !lead->copy_stream) {
    MH_HIP(hipStreamCreateWithFlags(&lead->copy_stream, hipStreamNonBlocking));
    MH_HIP(hipEventCreateWithFlags(&lead->ev_pairs_ready, hipEventDisableTiming));
    MH_HIP(hipEventCreateWithFlags(&lead->ev_pairs_copied, hipEventDisableTiming));
  }
  return MH_OK;
}

void set_pairs_fields(const PairsPlan& pp, size_t job_index, BatchJob& d) {
  if (!pp.want) return;
  d.cp_counts = pp.dev_hdr + pp.hdr_off[job_index];
  d.cp_out = reinterpret_cast<uint32_t*>(pp.dev_block + pp.off[job_index]);
  d.cp_stride = (uint32_t)(mh_pairs_block_bytes(d.n) / 24);
}

// compaction of every finished job's pairings into the block (descriptors `dj` already on the device) + the download
// `lead` owns the staging buffer, the copy stream and its events (plan_pairs: always the FIRST job's context, which is
// what mh_ctx_synchronize(scans[0]'s context) and the next batch wait on); `s` is the stream the compaction runs on --
// the lock-step group's, which is another context's when job 0 is trivial (same device: events order them).
mh_status finish_pairs(mh_ctx* lead, hipStream_t s, const PairsPlan& pp, const BatchJob* dj, uint32_t A, uint32_t gx_cov) {
  if (!pp.want || A == 0) return MH_OK;
  if (lead->pairs_copy_pending) MH_HIP(hipStreamWaitEvent(s, lead->ev_pairs_copied, 0));  // staging still being read
  hipLaunchKernelGGL(k_count_valid_b, dim3(gx_cov, A), dim3(kBlock), 0, s, dj);
  hipLaunchKernelGGL(k_scan_blocks_b, dim3(1, A), dim3(1024), 0, s, dj);
  hipLaunchKernelGGL(k_compact_b, dim3(gx_cov, A), dim3(kBlock), 0, s, dj);
  MH_HIP(hipGetLastError());
  if (pp.mem == MH_MEM_HOST) {
    MH_HIP(hipMemcpyAsync(pp.host_block, pp.dev_block, pp.total_bytes, hipMemcpyDeviceToHost, s));
    MH_HIP(mh::wait_stream(s));
  } else if (pp.mem == MH_MEM_HOST_PINNED) {
    // on the copy stream: the call returns, the next batch's kernels run while this block travels
    MH_HIP(hipEventRecord(lead->ev_pairs_ready, s));
    MH_HIP(hipStreamWaitEvent(lead->copy_stream, lead->ev_pairs_ready, 0));
    MH_HIP(hipMemcpyAsync(pp.host_block, pp.dev_block, pp.total_bytes, hipMemcpyDeviceToHost, lead->copy_stream));
    MH_HIP(hipEventRecord(lead->ev_pairs_copied, lead->copy_stream));
    lead->pairs_copy_pending = true;
  }
  return MH_OK;
}
}  // namespace

static mh_status align_batch_run(size_t n_jobs, const mh_map* const* maps, const mh_scan* const* scans,
                                 const mh_icp_params* params, int32_t params_per_job, const double* T_guesses,
                                 const mh_prior* const* priors, mh_icp_result* results, void* pairs_block, int32_t pairs_mem);

mh_status mh_icp_align_batch(size_t n_jobs, const mh_map* const* maps, const mh_scan* const* scans,
                             const mh_icp_params* params, int32_t params_per_job, const double* T_guesses,
                             const mh_prior* const* priors, mh_icp_result* results, void* pairs_block, int32_t pairs_mem) {
  const mh_status st = align_batch_run(n_jobs, maps, scans, params, params_per_job, T_guesses, priors, results, pairs_block, pairs_mem);
  // MH_DEBUG_VERIFY_BATCH=1 (development): every job once more as a single alignment -- a batch has to give the same bits
  if (st == MH_OK && n_jobs && getenv("MH_DEBUG_VERIFY_BATCH") != nullptr && !pairs_block) {
    for (size_t i = 0; i < n_jobs; i++) {
      mh_icp_result r2;
      const mh_icp_params* q = params_per_job ? &params[i] : params;
      if (mh_icp_align(maps[i], scans[i], q, T_guesses + 12 * i, priors ? priors[i] : nullptr, &r2, nullptr, nullptr, MH_MEM_HOST) != MH_OK) continue;
      if (memcmp(r2.T, results[i].T, sizeof(r2.T)) != 0 || r2.n_iterations != results[i].n_iterations) {
        double md = 0;
        for (int k = 0; k < 12; k++) md = fmax(md, fabs(r2.T[k] - results[i].T[k]));
        fprintf(stderr, "[MH_DEBUG_VERIFY_BATCH] job %zu of %zu: n = %zu points, batch %u iterations (reason %u, %u pairs) vs single %u (reason %u, %u pairs), max |dT| %.3e; sizes:",
                i, n_jobs, (size_t)scans[i]->n, results[i].n_iterations, results[i].termination_reason, results[i].n_final_pairs, r2.n_iterations,
                r2.termination_reason, r2.n_final_pairs, md);
        for (size_t k = 0; k < n_jobs; k++) fprintf(stderr, " %zu", (size_t)scans[k]->n);
        fprintf(stderr, "\n");
        // which of the two is unstable?  the batch once more, the single once more
        std::vector<mh_icp_result> again(n_jobs);
        if (align_batch_run(n_jobs, maps, scans, params, params_per_job, T_guesses, priors, again.data(), nullptr, pairs_mem) == MH_OK) {
          mh_icp_result r3;
          (void)mh_icp_align(maps[i], scans[i], q, T_guesses + 12 * i, priors ? priors[i] : nullptr, &r3, nullptr, nullptr, MH_MEM_HOST);
          fprintf(stderr, "[MH_DEBUG_VERIFY_BATCH]    second batch == first batch: %d, second batch == single: %d, second single == first single: %d (iterations %u / %u / %u / %u)\n",
                  memcmp(again[i].T, results[i].T, sizeof(r2.T)) == 0, memcmp(again[i].T, r2.T, sizeof(r2.T)) == 0,
                  memcmp(r3.T, r2.T, sizeof(r2.T)) == 0, results[i].n_iterations, again[i].n_iterations, r2.n_iterations, r3.n_iterations);
        }
      }
    }
  }
  return st;
}

static mh_status align_batch_run(size_t n_jobs, const mh_map* const* maps, const mh_scan* const* scans,
                                 const mh_icp_params* params, int32_t params_per_job, const double* T_guesses,
                                 const mh_prior* const* priors, mh_icp_result* results, void* pairs_block, int32_t pairs_mem) {
  MH_REQUIRE(n_jobs == 0 || (maps && scans && params && T_guesses && results), "null argument");
  MH_REQUIRE(!pairs_block || pairs_mem == MH_MEM_HOST || pairs_mem == MH_MEM_DEVICE || pairs_mem == MH_MEM_HOST_PINNED,
             "bad mem space");
  if (n_jobs == 0) return MH_OK;
  auto P = [&](size_t i) { return params_per_job ? &params[i] : params; };
  std::vector<AlignJob> jobs(n_jobs);
  for (size_t i = 0; i < n_jobs; i++) {
    MH_TRY(check_align_args(maps[i], scans[i], P(i), T_guesses + 12 * i, &results[i]));
    for (size_t j = 0; j < i; j++)
      MH_REQUIRE(scans[j]->ctx != scans[i]->ctx, "each job of a batch needs its own context");
    MH_REQUIRE(!pairs_block || scans[i]->ctx->device == scans[0]->ctx->device, "a pairs block needs all jobs on one device");
    jobs[i].defer_upload = n_jobs >= 2;  // a lock-step group uploads its jobs' blocks in one staged copy
    MH_TRY(jobs[i].start(maps[i], scans[i], P(i), T_guesses + 12 * i, priors ? priors[i] : nullptr, &results[i],
                         nullptr, i));
  }
  mh_ctx* lead0 = scans[0]->ctx;
  PairsPlan pp;
  MH_TRY(set_device(lead0));
  MH_TRY(plan_pairs(lead0, jobs, pairs_block, pairs_mem, pp));
  // Lock-step mode: every kernel of an iteration is ONE launch over all jobs of a group (blockIdx.y = job).  The jobs'
  // tails fill each other's idle lanes, which concurrent streams do not achieve (HIP maps them onto four hardware queues
  // whose kernels mostly run one after the other).  A group = the jobs that run the same kernel chain:
  //   quad / tile matcher + k_accum + k_solve (large layers), row matcher with the fused first accumulation (2-12 k
  //   points), row matcher + one-workgroup accumulate-and-solve (<= 2 k points: what lidar3d-default.yaml feeds), and the
  //   same with Matcher_Point2Plane riding along (lidar3d-ndt.yaml);
  // each job keeps its own parameters (iteration budget, schedules, hook check point, prior), state block, termination
  // flag and iteration count.  Jobs whose chain has no lock-step form (or that are alone in their group) take the
  // per-stream path below.
  enum Kind { K_NONE = 0, K_QUAD, K_TILE, K_WAVE, K_ORD, K_ROWF, K_STEP, K_STEP_PL, K_FLAT };
  const bool no_lockstep = getenv("MH_NO_LOCKSTEP") != nullptr;
  const bool batch_prof = !jobs.empty() && jobs[0].prof && !no_lockstep;  // (profile == 2 times job 0's share of a match kernel)
  auto kind_of = [&](const AlignJob& j) -> int {
    if (j.finished || no_lockstep || j.trace || j.prof) return K_NONE;
    // row-kernel layers up to 8 k points: k_step16_b, the chain of a single alignment with the jobs' workgroups side by side -- the
    // same sums in the same order, hence the same bits.  (Round 4 also kept the one-workgroup accumulate-and-solve of round 3 for
    // batches, re-ordered to give those bits: 4 / 8 / 16 sequences 3690 / 4920 / 5770 scans/s against 4092 / 5173 / 5510 this
    // way, NDT pipeline 3966 / 4509 / 3939 against 4780 / 5500 / 5600: removed.)
    if (j.use_step_chain() && !batch_prof) return j.pl ? K_STEP_PL : K_STEP;
    if (j.variant == 4 && !j.pl) return K_QUAD;
    if (j.variant == 9 && !j.pl) return K_FLAT;
    if (j.variant == 6 && !j.pl) return K_TILE;
    if (j.variant == 7 && !j.pl) return K_WAVE;
    if (j.variant == 8 && !j.pl) return K_ORD;
    if (j.variant == 5 && j.fused16 && !j.pl) return K_ROWF;
    return K_NONE;
  };
  struct Group {
    int kind = K_NONE;
    std::vector<AlignJob*> jobs;
    std::vector<size_t> index;
    mh_ctx* lead = nullptr;
    IcpDeviceState* h_states = nullptr;
    const BatchJob* dj = nullptr;
    uint32_t gx_match = 1, gx_acc = 1, gx_cov = 1, gx_step = 1, enq = 0, prof_n = 0, max_iterations = 0, inner = 1, chunk = 10;
    uint32_t src = 2, par = 0, launches = 0;  // k_step16_b: the state block (2 = canonical) and the partials half the next launch reads; launches so far
    bool cov = false, done = false, auto_chunk = false;
    bool loop_wave = false;  // ... as k_icpw_b (point layers): every job its own workgroups
    uint32_t gx_loopw = 1;   // ... whose launch has this many workgroups per job
    bool loop_now = false;   // k_icp16_b: the group's whole loops in ONE launch (decided below; cleared when a job's workgroups gave up)
    uint32_t loop_wgs = 0;   // ... and what it holds of the device's admission count meanwhile
    bool step_chain() const { return kind == K_STEP || kind == K_STEP_PL; }
    bool with_planes() const { return kind == K_STEP_PL; }
  };
  std::vector<Group> groups;
  struct ReleaseLoops {  // whatever way this call ends, the groups' share of the device's loop admission count is given back
    std::vector<Group>* gs;
    ~ReleaseLoops() {
      for (Group& g : *gs)
        if (g.loop_wgs && g.lead) {
          AlignJob::loop_count(g.lead->device).fetch_sub(g.loop_wgs);
          g.loop_wgs = 0;
        }
    }
  } release_loops{&groups};
  const bool want_prof = !jobs.empty() && jobs[0].prof && !no_lockstep;  // profile == 2: the share of job 0's group
  if (want_prof) jobs[0].prof = false;
  for (size_t i = 0; i < n_jobs; i++) {
    const int k = kind_of(jobs[i]);
    if (k == K_NONE) continue;
    const mh_icp_params* q = jobs[i].p;
    Group* g = nullptr;
    for (Group& c : groups)  // same chain, same device, same loop shape
      if (c.kind == k && c.lead->device == jobs[i].ctx->device && c.inner == q->gn.max_inner_iterations &&
          c.cov == (q->compute_covariance != 0))
        g = &c;
    if (!g) {
      groups.emplace_back();
      g = &groups.back();
      g->kind = k;
      g->lead = jobs[i].ctx;
      g->inner = q->gn.max_inner_iterations;
      g->cov = q->compute_covariance != 0;
      g->chunk = q->poll_every ? q->poll_every : 0;
      g->auto_chunk = q->poll_every == 0;
    }
    // automatic chunks: the group's first chunk is as long as its slowest job expects to run (every job's own estimate:
    // AlignJob::start) -- jobs that finish earlier leave their blocks at once, so only what lies beyond the LAST job's end is
    // wasted, while every poll in between drains the device for a host round trip (measured with fixed chunks of 10 on 8
    // sequences: 3.1 polls per alignment)
    if (g->auto_chunk && jobs[i].chunk > g->chunk) g->chunk = jobs[i].chunk;
    g->jobs.push_back(&jobs[i]);
    g->index.push_back(i);
    g->max_iterations = q->max_iterations > g->max_iterations ? q->max_iterations : g->max_iterations;
  }
  {  // a job alone in its group gains nothing from lock step; MH_LOCKSTEP_GROUPS splits the groups further (measured: two
     // groups of the C2 batch overlap one's match launch with the other's short launches for +3 %; default off)
    uint32_t split = 1;
    if (const char* e = getenv("MH_LOCKSTEP_GROUPS")) split = (uint32_t)atoi(e) > 0 ? (uint32_t)atoi(e) : 1u;
    std::vector<Group> kept;
    for (Group& g : groups) {
      if (g.jobs.size() < 2) continue;
      uint32_t parts = split;
      if (parts > g.jobs.size() / 2) parts = (uint32_t)(g.jobs.size() / 2);
      if (parts < 1 || pp.want) parts = 1;  // (the pairs block is compacted by one launch over one group's jobs)
      for (uint32_t part = 0; part < parts; part++) {
        Group h = g;
        h.jobs.clear();
        h.index.clear();
        for (size_t a = 0; a < g.jobs.size(); a++)
          if (a * parts / g.jobs.size() == part) {
            h.jobs.push_back(g.jobs[a]);
            h.index.push_back(g.index[a]);
          }
        h.lead = h.jobs[0]->ctx;
        kept.push_back(h);
      }
    }
    groups.swap(kept);
    if (groups.size() > 64) groups.resize(64);  // (their jobs fall through to the per-stream path)
  }
  std::vector<char> in_group(n_jobs, 0);
  for (Group& g : groups)
    for (size_t i : g.index) in_group[i] = 1;
  if (want_prof && !in_group[0]) jobs[0].prof = true;  // job 0 goes the per-stream way: its own events
  bool pairs_by_group = pp.want && groups.size() == 1 && groups[0].jobs.size() == (size_t)std::count_if(jobs.begin(), jobs.end(), [](const AlignJob& j) { return !j.finished; });

  if (!groups.empty()) {
    constexpr size_t kBlockBytes = kParamsOffset + sizeof(IcpDeviceParams);  // one job's [state | params] block
    static_assert(kBlockBytes % 4 == 0 && sizeof(BatchJob) % 8 == 0, "staging layout");
    for (Group& g : groups) {
      const uint32_t A = (uint32_t)g.jobs.size();
      mh_ctx* lead = g.lead;
      MH_TRY(set_device(lead));
      hipStream_t s = lead->stream;
#ifdef MH_DEV_VARIANTS
      if (g.kind == K_TILE || g.kind == K_WAVE)
        for (AlignJob* j : g.jobs) MH_TRY(scan_tiles_ready(j->scan));  // tile counts (the builds were queued by start())
#endif
      MH_TRY(order_after_job_streams(lead, g.jobs));
      size_t stage_bytes = 0;
      for (AlignJob* j : g.jobs) stage_bytes += kBlockBytes + j->nsched_pending * sizeof(double);
      const size_t need = A * sizeof(IcpDeviceState) + A * sizeof(BatchJob) + stage_bytes;
      MH_TRY(lead->batch_desc.reserve(A * sizeof(BatchJob) + stage_bytes));  // descriptors | staged blocks
      MH_TRY(lead->batch_states.reserve(A * sizeof(IcpDeviceState)));
      if (lead->h_batch_cap < need) {
        if (lead->h_batch) (void)hipHostFree(lead->h_batch);
        lead->h_batch = nullptr;
        lead->h_batch_cap = 0;
        MH_HIP(hipHostMalloc(&lead->h_batch, need, hipHostMallocDefault));
        lead->h_batch_cap = need;
      }
      g.h_states = reinterpret_cast<IcpDeviceState*>(lead->h_batch);
      BatchJob* h_desc = reinterpret_cast<BatchJob*>(reinterpret_cast<char*>(lead->h_batch) + A * sizeof(IcpDeviceState));
      char* h_stage = reinterpret_cast<char*>(h_desc) + A * sizeof(BatchJob);
      size_t off = 0;
      for (uint32_t a = 0; a < A; a++) {
        AlignJob& j = *g.jobs[a];
        BatchJob& d = h_desc[a];
        fill_batch_desc(j, d);
        if (pairs_by_group) set_pairs_fields(pp, g.index[a], d);
        d.stage_off = (uint32_t)(off / 4);
        uint32_t bm = (uint32_t)((4ull * d.n + kBlock - 1) / kBlock);  // quad
        if (g.kind == K_ROWF) bm = d.nbm;
        if (g.kind == K_FLAT) bm = nblk_flat(d.n);
        if (g.kind == K_TILE) bm = d.n_tiles;
        if (g.kind == K_WAVE) bm = d.n_tiles;
        if (g.step_chain()) {  // all jobs' workgroups resident at once: kStepMaxWorkgroups shared between them
          static const uint32_t cap_env = getenv("MH_STEP_WGS") ? (uint32_t)std::max(1, atoi(getenv("MH_STEP_WGS"))) : 0u;  // (development)
          const uint32_t cap = cap_env ? cap_env : (kStepMaxWorkgroups / A ? kStepMaxWorkgroups / A : 1u);
          const uint32_t ng = (d.n + kStepPoints - 1) / kStepPoints;
          const uint32_t nw = ng < cap ? ng : cap;
          g.gx_step = nw > g.gx_step ? nw : g.gx_step;
        }
        g.gx_match = bm > g.gx_match ? bm : g.gx_match;
        g.gx_acc = d.nba > g.gx_acc ? d.nba : g.gx_acc;
        g.gx_cov = d.nb > g.gx_cov ? d.nb : g.gx_cov;
        // this job's [state | params] mirror and its schedules into the staging area
        memcpy(h_stage + off, j.ctx->h_state, kBlockBytes);
        memcpy(h_stage + off + kBlockBytes, j.ctx->h_sched, j.nsched_pending * sizeof(double));
        off += kBlockBytes + j.nsched_pending * sizeof(double);
        j.defer_upload = false;
      }
      // Small layers: the whole loops of the group's jobs in ONE launch (k_icp16_b) when every job's layer has at most kLoopMaxGroups
      // groups, a workgroup takes at most kLoopGroupsPerWg of them, and the launch's workgroups are admitted (all resident together,
      // beside the one-launch loops of single alignments running on the device).  MH_NO_LOOP16 / MH_NO_LOOP16_BATCH: the chain.
      if (g.step_chain() && getenv("MH_NO_LOOP16") == nullptr && getenv("MH_NO_LOOP16_BATCH") == nullptr &&
          !AlignJob::loop_holdoff(lead->device, false)) {
        bool fits = true;
        // k_icpw_b (point layers): every job its own workgroups of 128 points; k_icp16_b (NDT maps, MH_NO_LOOPW): the jobs share
        // kStepMaxWorkgroups workgroups of 32 points, a workgroup taking several groups
        g.loop_wave = !g.with_planes() && loop_wave_enabled(true);
        uint32_t units = 0, gx_w = 1;
        for (uint32_t a = 0; a < A; a++) {
          const uint32_t ng = (h_desc[a].n + kStepPoints - 1) / kStepPoints;
          const uint32_t nw = ng < g.gx_step ? ng : g.gx_step;
          if (g.loop_wave) {
            fits = fits && ng <= kLwMaxGroups && g.jobs[a]->sk.max_iterations > 0;
            const uint32_t w = (ng + kLwGroups - 1) / kLwGroups;
            units += w;
            gx_w = w > gx_w ? w : gx_w;
          } else {
            fits = fits && ng <= kLoopMaxGroups && (nw == 0 || (ng + nw - 1) / nw <= kLoopGroupsPerWg) && g.jobs[a]->sk.max_iterations > 0;
          }
        }
        if (!g.loop_wave) units = kLoopUnitsPerCu * g.gx_step * A;
        if (fits && AlignJob::loop_admit(lead->device, units)) {
          g.loop_now = true;
          g.loop_wgs = units;
          g.gx_loopw = gx_w;
          for (uint32_t a = 0; a < A && g.loop_now; a++) {
            AlignJob& j = *g.jobs[a];
            if (g.loop_wave && (map_ensure_qidx(j.map, s) != MH_OK || !j.map->view().pts_q)) {  // (s waits for every job's stream: above)
              g.loop_now = false;
              break;
            }
            if (g.loop_wave) h_desc[a].map = j.map->view();  // (with the sub-voxel index)
            if (j.ctx->loop_x.bytes < kLoopExchangeBytes) {
              if (j.ctx->loop_x.reserve(kLoopExchangeBytes) != MH_OK) {
                g.loop_now = false;
                break;
              }
              (void)hipMemsetAsync(j.ctx->loop_x.p, 0, kLoopExchangeBytes, s);
            }
            const uint32_t max_steps = j.p->max_iterations * j.p->gn.max_inner_iterations + 1u;
            h_desc[a].loop_xa = j.ctx->loop_x.p;
            h_desc[a].loop_xb = static_cast<char*>(j.ctx->loop_x.p) + 2 * (size_t)kAccN * kLoopRowStride * 16;
            h_desc[a].loop_serial0 = j.ctx->loop_serial;
            h_desc[a].loop_pad = getenv("MH_LOOP16_TEST_ABANDON") ? 1u : 0u;
            j.ctx->loop_serial += max_steps + 2u;
          }
          if (!g.loop_now) {
            AlignJob::loop_count(lead->device).fetch_sub(g.loop_wgs);
            g.loop_wgs = 0;
          }
        }
      }
      // descriptors and staged blocks in ONE copy, then a scatter kernel writes every job's block where it lives
      MH_HIP(hipMemcpyAsync(lead->batch_desc.p, h_desc, A * sizeof(BatchJob) + stage_bytes, hipMemcpyHostToDevice, s));
      g.dj = lead->batch_desc.as<BatchJob>();
      hipLaunchKernelGGL(k_scatter_blocks, dim3(A), dim3(256), 0, s, g.dj,
                         reinterpret_cast<const uint32_t*>(lead->batch_desc.as<char>() + A * sizeof(BatchJob)),
                         (uint32_t)(kBlockBytes / 4));
    }
    // (Round 4 also built the streaming control of AlignJob::run_streaming for a whole lock-step group -- every job publishing in
    // its own progress word, the host following the slowest job still running -- and measured it on 4 / 8 / 16 sequences in one
    // process: 4073 / 4904 / 6597 scans/s against 4040 / 5255 / 6560 with the chunks below, NDT pipeline 4475 / 4963 / 5120
    // against 4186 / 5120 / 5245.  The spinning leader thread takes a core from the seven threads that queue uploads, filters
    // and map updates beside it, and a group's tail is amortised over its jobs anyway.  Removed; single alignments keep it.)
    for (;;) {
      bool any = false;
      uint32_t m_of[64] = {0};
      // enqueue one chunk per unfinished group, iteration by iteration across the groups so that their launches interleave
      uint32_t m_max = 0;
      for (size_t gi = 0; gi < groups.size(); gi++) {
        Group& g = groups[gi];
        if (g.done) continue;
        any = true;
        m_of[gi] = (g.max_iterations - g.enq) < g.chunk ? (g.max_iterations - g.enq) : g.chunk;
        if (g.loop_now) {  // everything in one launch, now
          m_of[gi] = g.max_iterations - g.enq;
          const uint32_t A = (uint32_t)g.jobs.size();
          g_loop16_runs.fetch_add(A);
          if (g.loop_wave) hipLaunchKernelGGL(k_icpw_b, dim3(g.gx_loopw, A), dim3(kLwThreads), 0, g.lead->stream, g.dj);
          else if (g.with_planes()) hipLaunchKernelGGL(k_icp16_b<true>, dim3(g.gx_step, A), dim3(kSolveThreads), 0, g.lead->stream, g.dj);
          else hipLaunchKernelGGL(k_icp16_b<false>, dim3(g.gx_step, A), dim3(kSolveThreads), 0, g.lead->stream, g.dj);
          continue;
        }
        m_max = m_of[gi] > m_max ? m_of[gi] : m_max;
      }
      if (!any) break;
      for (uint32_t it = 0; it < m_max; it++)
        for (size_t gi = 0; gi < groups.size(); gi++) {
          Group& g = groups[gi];
          if (g.done || g.loop_now || it >= m_of[gi]) continue;
          hipStream_t s = g.lead->stream;
          const uint32_t A = (uint32_t)g.jobs.size();
          const bool pr = want_prof && gi == 0 && g.jobs[0] == &jobs[0];
          if (g.step_chain()) {
            for (uint32_t in = 0; in < g.inner; in++) {
              if (g.with_planes()) hipLaunchKernelGGL(k_step16_b<true>, dim3(g.gx_step, A), dim3(kSolveThreads), 0, s, g.dj, g.src, g.par, 0u, g.launches);
              else hipLaunchKernelGGL(k_step16_b<false>, dim3(g.gx_step, A), dim3(kSolveThreads), 0, s, g.dj, g.src, g.par, 0u, g.launches);
              g.src = g.src == 2u ? 0u : (g.src ^ 1u);
              g.par ^= 1u;
              g.launches++;
            }
            continue;
          }
          if (pr) MH_HIP(hipEventRecord(g.lead->prof_ev[2 * g.prof_n], s));
          switch (g.kind) {
            case K_ROWF: hipLaunchKernelGGL(k_match16f_b, dim3(g.gx_match, A), dim3(kBlock), 0, s, g.dj); break;
#ifdef MH_DEV_VARIANTS
            case K_TILE: hipLaunchKernelGGL(k_match_tile_b, dim3(g.gx_match, A), dim3(kTileThreads), 0, s, g.dj); break;
            case K_WAVE:
              if (wave_lds_env()) hipLaunchKernelGGL(k_match_wave_dense_b<true>, dim3(g.gx_match, A), dim3(64), 0, s, g.dj);
              else hipLaunchKernelGGL(k_match_wave_dense_b<false>, dim3(g.gx_match, A), dim3(64), 0, s, g.dj);
              hipLaunchKernelGGL(k_match_wave_sparse_b, dim3(g.gx_match, A), dim3(kBlock), 0, s, g.dj);
              break;
            case K_ORD: hipLaunchKernelGGL(k_match4o_b, dim3(g.gx_match, A), dim3(kBlock), 0, s, g.dj); break;
#endif
            case K_FLAT: hipLaunchKernelGGL(k_match_flat_b, dim3(g.gx_match, A), dim3(kFlatThreads), 0, s, g.dj); break;
            default: hipLaunchKernelGGL(k_match4_b, dim3(g.gx_match, A), dim3(kBlock), 0, s, g.dj); break;
          }
          if (pr) {
            MH_HIP(hipEventRecord(g.lead->prof_ev[2 * g.prof_n + 1], s));
            g.prof_n++;
          }
          if (g.kind != K_ROWF) hipLaunchKernelGGL(g.kind == K_FLAT ? k_accum_b<true> : k_accum_b<false>, dim3(g.gx_acc, A), dim3(kBlock), 0, s, g.dj, 1u);
          hipLaunchKernelGGL(k_solve_b, dim3(1, A), dim3(kSolveThreads), 0, s, g.dj, 1u);
          for (uint32_t in = 1; in < g.inner; in++) {
            hipLaunchKernelGGL(g.kind == K_FLAT ? k_accum_b<true> : k_accum_b<false>, dim3(g.gx_acc, A), dim3(kBlock), 0, s, g.dj, 0u);
            hipLaunchKernelGGL(k_solve_b, dim3(1, A), dim3(kSolveThreads), 0, s, g.dj, 0u);
          }
        }
      for (size_t gi = 0; gi < groups.size(); gi++) {
        Group& g = groups[gi];
        if (g.done) continue;
        hipStream_t s = g.lead->stream;
        const uint32_t A = (uint32_t)g.jobs.size();
        if (g.step_chain() && !g.loop_now) {  // the pending Gauss-Newton step of every job, into the canonical state blocks
          if (g.with_planes()) hipLaunchKernelGGL(k_step16_b<true>, dim3(1, A), dim3(kSolveThreads), 0, s, g.dj, g.src, g.par, 1u, g.launches);
          else hipLaunchKernelGGL(k_step16_b<false>, dim3(1, A), dim3(kSolveThreads), 0, s, g.dj, g.src, g.par, 1u, g.launches);
          g.src = 2;
          g.launches++;
        }
        if (g.cov) {  // no-ops for jobs whose loop has not terminated
          hipLaunchKernelGGL(k_cov_prepare_b, dim3(1, A), dim3(64), 0, s, g.dj);
          hipLaunchKernelGGL(k_cov_accum_b, dim3(g.gx_cov, A), dim3(kBlock), 0, s, g.dj);
          if (g.with_planes()) hipLaunchKernelGGL(k_cov_accum_plbuf_b, dim3(g.gx_cov, A), dim3(kBlock), 0, s, g.dj);
          hipLaunchKernelGGL(k_cov_finalize_b, dim3(1, A), dim3(kSolveThreads), 0, s, g.dj);
        }
        hipLaunchKernelGGL(k_gather_states, dim3(A), dim3(256), 0, s, g.dj, g.lead->batch_states.as<IcpDeviceState>());
        MH_HIP(hipGetLastError());
        MH_HIP(hipMemcpyAsync(g.h_states, g.lead->batch_states.p, A * sizeof(IcpDeviceState), hipMemcpyDeviceToHost, s));
      }
      for (size_t gi = 0; gi < groups.size(); gi++) {
        Group& g = groups[gi];
        if (g.done) continue;
        const hipError_t we = mh::wait_stream(g.lead->stream);
        if (g.loop_wgs) {
          AlignJob::loop_count(g.lead->device).fetch_sub(g.loop_wgs);
          g.loop_wgs = 0;
        }
        MH_HIP(we);
        if (g.loop_now) {
          // a job whose workgroups gave up waiting for each other left done == 0 and its canonical state block as uploaded: the
          // group goes on launch by launch (k_step16_b from the start; the jobs that did finish are no-ops there)
          g.loop_now = false;
          bool abandoned = false;
          for (size_t a = 0; a < g.jobs.size(); a++) {
            const IcpDeviceState& h = g.h_states[a];
            if (g.jobs[a]->finished || (h.done && !h.handover_timeouts)) continue;
            abandoned = true;
            g_loop16_fallbacks.fetch_add(1);
            MH_HIP(hipMemsetAsync(&g.jobs[a]->ctx->d_state->handover_timeouts, 0, sizeof(uint32_t) * 10, g.lead->stream));
          }
          if (abandoned) {
            if (getenv("MH_LOOP16_TEST_ABANDON") == nullptr) AlignJob::loop_holdoff(g.lead->device, true);
            for (size_t a = 0; a < g.jobs.size(); a++) {
              AlignJob& j = *g.jobs[a];
              const IcpDeviceState& h = g.h_states[a];
              if (j.finished || !(h.done && !h.handover_timeouts)) continue;
              memcpy(j.ctx->h_state, &h, sizeof(IcpDeviceState));
              j.enqueued = j.p->max_iterations;
              MH_TRY(j.poll(true));
            }
            g.enq = 0;
            g.src = 2;
            g.par = 0;
            g.launches = 0;
            continue;  // (not done: the next round enqueues the chain's first chunk)
          }
        }
        g.enq += m_of[gi];
        g.done = true;
        if (g.auto_chunk) g.chunk = 8;  // follow-up chunks
        for (size_t a = 0; a < g.jobs.size(); a++) {
          AlignJob& j = *g.jobs[a];
          if (j.finished) continue;
          memcpy(j.ctx->h_state, &g.h_states[a], sizeof(IcpDeviceState));
          j.enqueued = g.enq < j.p->max_iterations ? g.enq : j.p->max_iterations;
          MH_TRY(j.poll(true));
          g.done = g.done && j.finished;
        }
      }
    }
    if (want_prof && groups[0].jobs[0] == &jobs[0]) {  // the match step of job 0 = its share of its group's lock-step launches
      Group& g = groups[0];
      float ms = 0.f;
      double sum = 0.0;
      const mh_icp_result* r0 = jobs[0].res;
      uint32_t live = r0->n_iterations + ((r0->termination_reason == MH_TERM_MAX_ITERATIONS) ? 0u : 1u);
      if (live > g.prof_n) live = g.prof_n;
      for (uint32_t i = 0; i < live; i++) {
        MH_HIP(hipEventElapsedTime(&ms, g.lead->prof_ev[2 * i], g.lead->prof_ev[2 * i + 1]));
        sum += ms;
      }
      jobs[0].res->n_match_launches = live;
      jobs[0].res->match_kernel_ms = sum / (double)g.jobs.size();
      jobs[0].res->total_ms = 0.0;
    }
    if (pairs_by_group) {
      MH_TRY(finish_pairs(lead0, groups[0].lead->stream, pp, groups[0].dj, (uint32_t)groups[0].jobs.size(), groups[0].gx_cov));
      return MH_OK;
    }
  }
  // everything that is not in a lock-step group: one stream per job, chunks enqueued round robin
  for (size_t i = 0; i < n_jobs; i++)
    if (!in_group[i]) MH_TRY(jobs[i].flush_deferred());
  for (;;) {
    bool any = false;
    for (size_t i = 0; i < n_jobs; i++)
      if (!in_group[i] && !jobs[i].finished) {
        MH_TRY(jobs[i].enqueue_chunk());
        any = true;
      }
    if (!any) break;
    for (size_t i = 0; i < n_jobs; i++)
      if (!in_group[i]) MH_TRY(jobs[i].poll());
  }
  if (pp.want) {
    // every job has terminated and its stream is drained: one compaction launch over all of them on the first job's stream
    std::vector<size_t> act;
    for (size_t i = 0; i < n_jobs; i++)
      if (!jobs[i].trivial) act.push_back(i);
    if (act.empty()) return MH_OK;
    mh_ctx* lead = lead0;
    MH_TRY(set_device(lead));
    const uint32_t A = (uint32_t)act.size();
    MH_TRY(lead->batch_desc.reserve(A * sizeof(BatchJob)));
    if (lead->h_batch_cap < A * sizeof(BatchJob)) {
      if (lead->h_batch) (void)hipHostFree(lead->h_batch);
      lead->h_batch = nullptr;
      lead->h_batch_cap = 0;
      MH_HIP(hipHostMalloc(&lead->h_batch, A * sizeof(BatchJob), hipHostMallocDefault));
      lead->h_batch_cap = A * sizeof(BatchJob);
    }
    BatchJob* h_desc = reinterpret_cast<BatchJob*>(lead->h_batch);
    uint32_t gx_cov = 1;
    for (uint32_t a = 0; a < A; a++) {
      fill_batch_desc(jobs[act[a]], h_desc[a]);
      set_pairs_fields(pp, act[a], h_desc[a]);
      gx_cov = h_desc[a].nb > gx_cov ? h_desc[a].nb : gx_cov;
    }
    MH_HIP(hipMemcpyAsync(lead->batch_desc.p, h_desc, A * sizeof(BatchJob), hipMemcpyHostToDevice, lead->stream));
    MH_TRY(finish_pairs(lead, lead->stream, pp, lead->batch_desc.as<BatchJob>(), A, gx_cov));
    if (pp.mem != MH_MEM_HOST) MH_HIP(mh::wait_stream(lead->stream));  // h_batch is reused by the next batch
  }
  return MH_OK;
}

namespace {
#ifdef MH_DEV_VARIANTS
// MH_MATCH=t: the matcher-granular entry points run the tile matcher too (the parity tests drive every search kernel
// through mh_nn_search / mh_nn_search_dense); thr2 = +inf: no threshold
mh_status launch_tile_search(const mh_map* map, const mh_scan* scan, const double T[12], float thr2, float ang2) {
  mh_ctx* ctx = scan->ctx;
  const bool wave = tile_points_for_env() == 64u;
  MH_TRY(scan_build_tiles(scan, map->inv_vs, wave ? 64u : 256u));
  MH_TRY(scan_tiles_ready(scan));
  MH_TRY(map_ensure_qidx(map, ctx->stream));  // (sparse tiles are searched by quads)
  MH_HIP(mh::wait_stream(ctx->stream));  // the pinned state mirror may still be travelling
  init_state(ctx->h_state, T);
  ctx->h_state->cur_thr2 = thr2;
  ctx->h_state->cur_ang2 = ang2;
  MH_HIP(hipMemcpyAsync(ctx->d_state, ctx->h_state, sizeof(IcpDeviceState), hipMemcpyHostToDevice, ctx->stream));
  if (scan->n_tiles && wave)
    MH_LAUNCH_WAVE(ctx->stream, ctx->d_state, scan, map->view(), ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(), MH_WT_NULL);
  else if (scan->n_tiles)
    hipLaunchKernelGGL(k_match_tile, dim3(scan->n_tiles), dim3(kTileThreads), 0, ctx->stream, ctx->d_state, scan->sx, scan->sy,
                       scan->sz, scan->perm, scan->tile_start, scan->n_tiles, map->view(), ctx->pair_q.as<float4>(),
                       ctx->pair_gidx.as<uint32_t>()
#ifdef MH_DEBUG_WAVETRACE
                       , (unsigned long long*)nullptr
#endif
    );
  MH_HIP(hipGetLastError());
  return MH_OK;
}
inline bool tile_search_forced() {
  const char* e = getenv("MH_MATCH");
  return e && (e[0] == 't' || e[0] == 'w');
}
#else  // the shipped library: the matcher-granular entry points run k_match<false, 1> whatever MH_MATCH says
inline bool tile_search_forced() { return false; }
inline mh_status launch_tile_search(const mh_map*, const mh_scan*, const double*, float, float) { return MH_OK; }
#endif
}  // namespace

mh_status mh_nn_search(const mh_map* map, const mh_scan* scan, const double T[12], double threshold,
                       double threshold_angular_deg, const mh_pairs_out* out, int32_t mem, mh_match_info* info) {
  MH_REQUIRE(map && scan && T, "null argument");
  MH_REQUIRE(mem == MH_MEM_HOST || mem == MH_MEM_DEVICE, "bad mem space");
  MH_REQUIRE(map->ctx->device == scan->ctx->device, "map and scan live on different devices");
  MH_REQUIRE(pose_ok(T), "non-finite pose");
  mh_ctx* ctx = scan->ctx;
  MH_TRY(set_device(ctx));
  MH_TRY(map_ready_on(map, ctx->stream));
  if (info) {
    info->n_pairs = 0;
    info->potential_pairings = scan->n;  // counted before any test (App.B U6)
  }
  if (scan->n == 0) return MH_OK;
  MH_TRY(ensure_state(ctx));
  MH_TRY(ensure_pair_buffers(ctx, scan->n));
  PoseArg Ta;
  for (int i = 0; i < 12; i++) Ta.m[i] = T[i];
  MatchK mk{};
  SolveK sk0{};
  const double ang = threshold_angular_deg * 3.14159265358979323846 / 180.0;
  mk.ang2 = (float)(ang * ang);
  MH_TRY(upload_params(ctx, mk, sk0));
  if (tile_search_forced())
    MH_TRY(launch_tile_search(map, scan, T, (float)(threshold * threshold), mk.ang2));
  else
    hipLaunchKernelGGL((k_match<false, 1>), dim3(nblk(scan->n)), dim3(kBlock), 0, ctx->stream, ctx->d_state, Ta,
                       (float)(threshold * threshold), 1u, &ctx->d_params->mk, scan->x, scan->y, scan->z, (uint32_t)scan->n, map->view(),
                       ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(), (double*)nullptr, 0u);
  MH_HIP(hipGetLastError());
  mh_pairs_out none{};
  uint64_t np = 0;
  MH_TRY(compact_pairs(ctx, scan->n, out ? out : &none, mem, &np));
  if (info) info->n_pairs = np;
  return MH_OK;
}

mh_status mh_nn_search_k(const mh_map* map, const mh_scan* scan, const double T[12], double threshold,
                         double threshold_angular_deg, uint32_t pairings_per_point, const mh_pairs_out* out, int32_t mem,
                         mh_match_info* info) {
  MH_REQUIRE(map && scan && T, "null argument");
  MH_REQUIRE(mem == MH_MEM_HOST || mem == MH_MEM_DEVICE, "bad mem space");
  MH_REQUIRE(pairings_per_point >= 1 && pairings_per_point <= (uint32_t)kMaxKnn, "pairings_per_point must be 1..MH_MAX_PAIRINGS_PER_POINT");
  MH_REQUIRE(map->ctx->device == scan->ctx->device, "map and scan live on different devices");
  MH_REQUIRE(pose_ok(T), "non-finite pose");
  MH_REQUIRE((uint64_t)scan->n * pairings_per_point < 0xFFFFFFFFull, "scan size * pairings_per_point does not fit 32 bits");
  if (pairings_per_point == 1) return mh_nn_search(map, scan, T, threshold, threshold_angular_deg, out, mem, info);
  mh_ctx* ctx = scan->ctx;
  MH_TRY(set_device(ctx));
  MH_TRY(map_ready_on(map, ctx->stream));
  const uint32_t k = pairings_per_point;
  if (info) {
    info->n_pairs = 0;
    info->potential_pairings = (uint64_t)scan->n * k;  // pcLocal.size() * pairingsPerPoint, counted before any test (App.B U6)
  }
  if (scan->n == 0) return MH_OK;
  const size_t nk = scan->n * (size_t)k;
  MH_TRY(ensure_pair_buffers(ctx, nk));
  PoseArg Ta;
  for (int i = 0; i < 12; i++) Ta.m[i] = T[i];
  const double ang = threshold_angular_deg * 3.14159265358979323846 / 180.0;
  hipLaunchKernelGGL(k_match_kbest, dim3(nblk(scan->n)), dim3(kBlock), 0, ctx->stream, Ta, (float)(threshold * threshold),
                     (float)(ang * ang), k, scan->x, scan->y, scan->z, (uint32_t)scan->n, map->view(), ctx->pair_q.as<float4>(),
                     ctx->pair_gidx.as<uint32_t>());
  MH_HIP(hipGetLastError());
  mh_pairs_out none{};
  uint64_t np = 0;
  MH_TRY(compact_pairs(ctx, nk, out ? out : &none, mem, &np));
  // compact_pairs numbers the ENTRIES: entry e belongs to local point e / k
  if (out && out->local_idx && np) {
    if (mem == MH_MEM_HOST) {
      for (uint64_t e = 0; e < np; e++) out->local_idx[e] /= k;
    } else {
      hipLaunchKernelGGL(k_div_idx, dim3(nblk(np)), dim3(kBlock), 0, ctx->stream, out->local_idx, (uint32_t)np, k);
      MH_HIP(hipGetLastError());
      MH_HIP(mh::wait_stream(ctx->stream));
    }
  }
  if (info) info->n_pairs = np;
  return MH_OK;
}

mh_status mh_nn_search_dense(const mh_map* map, const mh_scan* scan, const double T[12], uint32_t* global_idx, float* gx,
                             float* gy, float* gz, float* d2, int32_t mem) {
  MH_REQUIRE(map && scan && T, "null argument");
  MH_REQUIRE(mem == MH_MEM_HOST || mem == MH_MEM_DEVICE, "bad mem space");
  MH_REQUIRE(map->ctx->device == scan->ctx->device, "map and scan live on different devices");
  MH_REQUIRE(pose_ok(T), "non-finite pose");
  mh_ctx* ctx = scan->ctx;
  MH_TRY(set_device(ctx));
  MH_TRY(map_ready_on(map, ctx->stream));
  const size_t n = scan->n;
  if (n == 0) return MH_OK;
  MH_TRY(ensure_state(ctx));
  MH_TRY(ensure_pair_buffers(ctx, n));
  PoseArg Ta;
  for (int i = 0; i < 12; i++) Ta.m[i] = T[i];
  MatchK mk{};
  SolveK sk0{};
  hipStream_t s = ctx->stream;
  MH_TRY(upload_params(ctx, mk, sk0));
  if (tile_search_forced())
    MH_TRY(launch_tile_search(map, scan, T, __builtin_inff(), 0.f));
  else
    hipLaunchKernelGGL((k_match<false, 1>), dim3(nblk(n)), dim3(kBlock), 0, s, ctx->d_state, Ta, 0.f, 0u,
                       &ctx->d_params->mk, scan->x,
                       scan->y, scan->z, (uint32_t)n, map->view(), ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(),
                       (double*)nullptr, 0u);
  uint32_t* o_gi = global_idx;
  float *o_x = gx, *o_y = gy, *o_z = gz, *o_d2 = d2;
  const size_t n4 = ((n + 63) / 64) * 64;
  if (mem == MH_MEM_HOST) {
    MH_TRY(ctx->compact.reserve(5 * n4 * 4));
    char* st = ctx->compact.as<char>();
    o_gi = global_idx ? (uint32_t*)st : nullptr;
    o_x = gx ? (float*)(st + n4 * 4) : nullptr;
    o_y = gy ? (float*)(st + 2 * n4 * 4) : nullptr;
    o_z = gz ? (float*)(st + 3 * n4 * 4) : nullptr;
    o_d2 = d2 ? (float*)(st + 4 * n4 * 4) : nullptr;
  }
  hipLaunchKernelGGL(k_unpack_dense, dim3(nblk(n)), dim3(kBlock), 0, s, ctx->pair_gidx.as<uint32_t>(),
                     ctx->pair_q.as<float4>(), (uint32_t)n, o_gi, o_x, o_y, o_z, o_d2);
  MH_HIP(hipGetLastError());
  MH_HIP(mh::wait_stream(s));
  if (mem == MH_MEM_HOST) {
    if (global_idx) MH_HIP(hipMemcpy(global_idx, o_gi, n * 4, hipMemcpyDeviceToHost));
    if (gx) MH_HIP(hipMemcpy(gx, o_x, n * 4, hipMemcpyDeviceToHost));
    if (gy) MH_HIP(hipMemcpy(gy, o_y, n * 4, hipMemcpyDeviceToHost));
    if (gz) MH_HIP(hipMemcpy(gz, o_z, n * 4, hipMemcpyDeviceToHost));
    if (d2) MH_HIP(hipMemcpy(d2, o_d2, n * 4, hipMemcpyDeviceToHost));
  }
  return MH_OK;
}

// compaction of the context's point-to-plane pairing buffers into caller arrays
static mh_status compact_pl_pairs(mh_ctx* ctx, size_t n, const mh_pairs_pl_out* out, int32_t mem, uint64_t* n_pairs_out) {
  hipStream_t s = ctx->stream;
  const uint32_t nb = nblk(n);
  const size_t n4 = ((n + 63) / 64) * 64;
  // layout: flags[n4] | counts[nb] | offsets[nb] | total[1] | (host staging) li,cx,cy,cz,nx,ny,nz [n4 each]
  const size_t hdr = ((n4 + (size_t)2 * nb + 1) * 4 + 255) / 256 * 256;
  MH_TRY(ctx->compact.reserve(hdr + 7 * n4 * 4));
  uint32_t* flags = ctx->compact.as<uint32_t>();
  uint32_t* counts = flags + n4;
  uint32_t* offsets = counts + nb;
  uint32_t* total = offsets + nb;
  char* stage = ctx->compact.as<char>() + hdr;
  void* o[7] = {out->local_idx, out->cx, out->cy, out->cz, out->nx, out->ny, out->nz};
  void* d[7];
  for (int a = 0; a < 7; a++) d[a] = (mem == MH_MEM_DEVICE) ? o[a] : (o[a] ? (void*)(stage + (size_t)a * n4 * 4) : nullptr);
  uint32_t h_total = 0;
  if (n) {
    hipLaunchKernelGGL(k_pl_flags, dim3(nb), dim3(kBlock), 0, s, ctx->pl_c.as<float4>(), (uint32_t)n, flags);
    hipLaunchKernelGGL(k_count_valid, dim3(nb), dim3(kBlock), 0, s, flags, (uint32_t)n, counts);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, s, counts, nb, offsets, total);
    hipLaunchKernelGGL(k_compact_pl, dim3(nb), dim3(kBlock), 0, s, flags, ctx->pl_c.as<float4>(), ctx->pl_n.as<float4>(),
                       (uint32_t)n, offsets, (uint32_t*)d[0], (float*)d[1], (float*)d[2], (float*)d[3], (float*)d[4],
                       (float*)d[5], (float*)d[6]);
    MH_HIP(hipGetLastError());
    MH_HIP(hipMemcpyAsync(&h_total, total, 4, hipMemcpyDeviceToHost, s));
    MH_HIP(mh::wait_stream(s));
  }
  if (mem == MH_MEM_HOST && h_total)
    for (int a = 0; a < 7; a++)
      if (o[a]) MH_HIP(hipMemcpy(o[a], d[a], (size_t)h_total * 4, hipMemcpyDeviceToHost));
  if (n_pairs_out) *n_pairs_out = h_total;
  return MH_OK;
}

mh_status mh_nn_search_pt2pl(const mh_map* map, const mh_scan* scan, const double T[12], double distance_threshold,
                             uint32_t mode, const mh_pairs_pl_out* out, int32_t mem, mh_match_info* info) {
  MH_REQUIRE(map && scan && T, "null argument");
  MH_REQUIRE(mode == MH_PT2PL_PLANE_DISTANCE || mode == MH_PT2PL_CENTROID_DISTANCE, "bad pt2pl mode");
  distance_threshold = (mode == MH_PT2PL_CENTROID_DISTANCE ? -1.0 : 1.0) * fabs(distance_threshold);
  MH_REQUIRE(mem == MH_MEM_HOST || mem == MH_MEM_DEVICE, "bad mem space");
  MH_REQUIRE(map->ctx->device == scan->ctx->device, "map and scan live on different devices");
  MH_REQUIRE(pose_ok(T), "non-finite pose");
  MH_REQUIRE(map->view().ndt, "the map carries no NDT statistics (build it with ndt_max_eigen_ratio > 0)");
  mh_ctx* ctx = scan->ctx;
  MH_TRY(set_device(ctx));
  MH_TRY(map_ready_on(map, ctx->stream));
  if (info) {
    info->n_pairs = 0;
    info->potential_pairings = scan->n;
  }
  if (scan->n == 0) return MH_OK;
  MH_TRY(ensure_state(ctx));
  MH_TRY(ensure_pl_buffers(ctx, scan->n));
  PoseArg Ta;
  for (int i = 0; i < 12; i++) Ta.m[i] = T[i];
  MatchK mk{};
  SolveK sk0{};
  MH_TRY(upload_params(ctx, mk, sk0));
  hipLaunchKernelGGL(k_match_pl<false>, dim3(nblk(scan->n)), dim3(kBlock), 0, ctx->stream, ctx->d_state, Ta,
                     (float)distance_threshold, &ctx->d_params->mk, scan->x, scan->y, scan->z, (uint32_t)scan->n, map->view(),
                     ctx->pl_c.as<float4>(), ctx->pl_n.as<float4>(), (double*)nullptr, 0u);
  MH_HIP(hipGetLastError());
  mh_pairs_pl_out none{};
  uint64_t np = 0;
  MH_TRY(compact_pl_pairs(ctx, scan->n, out ? out : &none, mem, &np));
  if (info) info->n_pairs = np;
  return MH_OK;
}

mh_status mh_nn_search_pt2pl_knn(const mh_map* map, const mh_scan* scan, const double T[12], const mh_pt2pl_knn_params* params,
                                 const mh_pairs_pl_out* out, int32_t mem, mh_match_info* info) {
  MH_REQUIRE(map && scan && T && params, "null argument");
  MH_REQUIRE(mem == MH_MEM_HOST || mem == MH_MEM_DEVICE, "bad mem space");
  MH_REQUIRE(map->ctx->device == scan->ctx->device, "map and scan live on different devices");
  MH_REQUIRE(pose_ok(T), "non-finite pose");
  MH_REQUIRE(params->knn >= 3 && params->knn <= (uint32_t)kMaxPlaneKnn, "knn must be 3..MH_MAX_PLANE_KNN");
  MH_REQUIRE(isfinite(params->distance_threshold) && isfinite(params->plane_eigen_threshold) && isfinite(params->search_radius) &&
             params->search_radius > 0.0, "bad thresholds");
  mh_ctx* ctx = scan->ctx;
  MH_TRY(set_device(ctx));
  MH_TRY(map_ready_on(map, ctx->stream));
  if (info) {
    info->n_pairs = 0;
    info->potential_pairings = scan->n;
  }
  if (scan->n == 0) return MH_OK;
  MH_TRY(ensure_pl_buffers(ctx, scan->n));
  PoseArg Ta;
  for (int i = 0; i < 12; i++) Ta.m[i] = T[i];
  PlKnnArg a;
  a.distance_threshold = params->distance_threshold;
  a.plane_eigen_threshold = params->plane_eigen_threshold;
  a.radius2 = (float)(params->search_radius * params->search_radius);
  a.knn = params->knn;
  a.min_pts = params->minimum_plane_points < 3u ? 3u : params->minimum_plane_points;  // (three points span a plane)
  hipLaunchKernelGGL(k_match_pl_knn, dim3(nblk(scan->n)), dim3(kBlock), 0, ctx->stream, Ta, a, scan->x, scan->y, scan->z,
                     (uint32_t)scan->n, map->view(), ctx->pl_c.as<float4>(), ctx->pl_n.as<float4>());
  MH_HIP(hipGetLastError());
  mh_pairs_pl_out none{};
  uint64_t np = 0;
  MH_TRY(compact_pl_pairs(ctx, scan->n, out ? out : &none, mem, &np));
  if (info) info->n_pairs = np;
  return MH_OK;
}

mh_status mh_icp_get_pt2pl_pairs(const mh_scan* scan, const mh_pairs_pl_out* out, int32_t mem, uint64_t* n_pairs) {
  MH_REQUIRE(scan && out, "null argument");
  MH_REQUIRE(mem == MH_MEM_HOST || mem == MH_MEM_DEVICE, "bad mem space");
  mh_ctx* ctx = scan->ctx;
  MH_TRY(set_device(ctx));
  if (n_pairs) *n_pairs = 0;
  if (scan->n == 0 || !ctx->pl_c.p || ctx->pl_c.bytes < scan->n * sizeof(float4)) return MH_OK;  // no pt2pl matcher has run
  return compact_pl_pairs(ctx, scan->n, out, mem, n_pairs);
}

// ---- solver-granular entry points ---------------------------------------------------------------
namespace {
// stage 3 (or 6/9) SoA float arrays of n elements into one device buffer with a common stride
mh_status stage_soa(mh_ctx* ctx, DevBuf& buf, const float* const* arrs, int count, size_t n, int32_t mem, size_t* stride_out) {
  const size_t stride = ((n + 63) / 64) * 64;
  MH_TRY(buf.reserve((size_t)count * stride * sizeof(float) + 256));
  for (int a = 0; a < count; a++)
    MH_TRY(stage_in(ctx, buf, (size_t)a * stride * sizeof(float), arrs[a], n * sizeof(float), mem));
  *stride_out = stride;
  return MH_OK;
}
}  // namespace

mh_status mh_gn_solve(mh_ctx* ctx, const mh_pairs_pt2pt* pp, const mh_pairs_pt2pl* pl, int32_t mem,
                      const mh_gn_params* p, const mh_prior* prior, double T_io[12], int32_t* n_steps, int32_t* solver_ok,
                      mh_gn_step* trace) {
  MH_REQUIRE(ctx && p && T_io, "null argument");
  MH_REQUIRE(mem == MH_MEM_HOST || mem == MH_MEM_DEVICE, "bad mem space");
  MH_REQUIRE(p->max_inner_iterations >= 1 && p->max_inner_iterations <= kMaxGnTrace, "max_inner_iterations out of [1,16]");
  MH_REQUIRE(p->robust_kernel <= MH_KERNEL_GM_C2, "unknown robust kernel");
  MH_REQUIRE(pose_ok(T_io), "non-finite linearisation point");
  const size_t np = pp ? pp->n : 0, nl = pl ? pl->n : 0;
  MH_REQUIRE(np == 0 || (pp->lx && pp->ly && pp->lz && pp->gx && pp->gy && pp->gz), "null pt2pt arrays");
  MH_REQUIRE(nl == 0 || (pl->lx && pl->ly && pl->lz && pl->cx && pl->cy && pl->cz && pl->nx && pl->ny && pl->nz),
             "null pt2pl arrays");
  if (n_steps) *n_steps = 0;
  if (solver_ok) *solver_ok = 1;
  MH_TRY(set_device(ctx));
  MH_TRY(ensure_state(ctx));
  hipStream_t s = ctx->stream;
  MH_HIP(mh::wait_stream(s));
  // stage pairings: build_a = pt2pt (l xyz | g xyz), build_b = pt2pl (l | c | n)
  size_t sp = 0, sl = 0;
  if (np) {
    const float* arrs[6] = {pp->lx, pp->ly, pp->lz, pp->gx, pp->gy, pp->gz};
    MH_TRY(stage_soa(ctx, ctx->build_a, arrs, 6, np, mem, &sp));
  }
  if (nl) {
    const float* arrs[9] = {pl->lx, pl->ly, pl->lz, pl->cx, pl->cy, pl->cz, pl->nx, pl->ny, pl->nz};
    MH_TRY(stage_soa(ctx, ctx->build_b, arrs, 9, nl, mem, &sl));
  }
  MH_TRY(ensure_pair_buffers(ctx, np));
  const uint32_t nbp = np ? nblk(np) : 0, nbl = nl ? nblk(nl) : 0;
  MH_TRY(ctx->partials_b.reserve((size_t)kGenN * (nbl ? nbl : 1) * sizeof(double)));
  MH_TRY(ctx->trace.reserve(sizeof(mh_gn_step) * kMaxGnTrace));
  const float* L = ctx->build_a.as<float>();
  if (np)
    hipLaunchKernelGGL(k_pack_pairs, dim3(nbp), dim3(kBlock), 0, s, L + 3 * sp, (uint32_t)np, (uint32_t)sp,
                       ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>());
  init_state(ctx->h_state, T_io);
  ctx->h_state->cur_kparam = p->robust_kernel_param;  // solver-granular path: fixed robust-kernel parameter
  MH_HIP(hipMemcpyAsync(ctx->d_state, ctx->h_state, sizeof(IcpDeviceState), hipMemcpyHostToDevice, s));
  MatchK mk{};
  mk.kernel = p->robust_kernel;
  mk.w_pt2pt = p->weight_pt2pt;
  SolveK sk;
  memset(&sk, 0, sizeof(sk));
  sk.max_iterations = 1;
  sk.disable_stall = 1;
  sk.max_inner = p->max_inner_iterations;
  sk.min_delta = p->min_delta;
  sk.max_cost = p->max_cost;
  fill_prior(sk, prior);
  sk.gn_trace = (mh_gn_step*)ctx->trace.p;
  mk.use_fixed = 1;
  mk.kparam_fixed = p->robust_kernel_param;
  MH_TRY(upload_params(ctx, mk, sk));
  MH_HIP(hipMemsetAsync(ctx->trace.p, 0, sizeof(mh_gn_step) * kMaxGnTrace, s));
  const float* P = ctx->build_b.as<float>();
  for (uint32_t in = 0; in < p->max_inner_iterations; in++) {
    if (np)
      hipLaunchKernelGGL(k_accum<false>, dim3(nblk_acc(np)), dim3(kBlock), 0, s, ctx->d_state, in == 0 ? 1u : 0u, &ctx->d_params->mk,
                         L, L + sp, L + 2 * sp, (uint32_t)np, ctx->pair_q.as<float4>(),
                         ctx->pair_gidx.as<uint32_t>(), ctx->partials.as<double>(), nblk_acc(np));
    if (nl)
      hipLaunchKernelGGL(k_accum_pl, dim3(nbl), dim3(kBlock), 0, s, ctx->d_state, p->robust_kernel,
                         p->robust_kernel_param, p->weight_pt2pl, P, P + 3 * sl, P + 6 * sl, (uint32_t)nl, (uint32_t)sl,
                         ctx->partials_b.as<double>(), nbl);
    hipLaunchKernelGGL(k_solve, dim3(1), dim3(kSolveThreads), 0, s, ctx->d_state, &ctx->d_params->sk,
                       ctx->partials.as<double>(), np ? nblk_acc(np) : 0u, np ? nblk_acc(np) : 0u,
                       ctx->partials_b.as<double>(), nbl, nbl, in == 0 ? 1u : 0u);
  }
  MH_HIP(hipGetLastError());
  MH_HIP(hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(IcpDeviceState), hipMemcpyDeviceToHost, s));
  MH_HIP(mh::wait_stream(s));
  const IcpDeviceState* h = ctx->h_state;
  for (int i = 0; i < 12; i++) T_io[i] = h->T[i];
  if (n_steps) *n_steps = (int32_t)h->n_solves;
  if (solver_ok) *solver_ok = (int32_t)h->solver_ok;
  if (trace) MH_HIP(hipMemcpy(trace, ctx->trace.p, sizeof(mh_gn_step) * p->max_inner_iterations, hipMemcpyDeviceToHost));
  return MH_OK;
}

mh_status mh_covariance(mh_ctx* ctx, const mh_pairs_pt2pt* pp, const mh_pairs_pt2pl* pl, int32_t mem, const double T[12],
                        double findif_xyz, double findif_ang, double cov[36]) {
  MH_REQUIRE(ctx && T && cov, "null argument");
  MH_REQUIRE(mem == MH_MEM_HOST || mem == MH_MEM_DEVICE, "bad mem space");
  MH_REQUIRE(findif_xyz > 0 && findif_ang > 0, "finite-difference steps must be > 0");
  MH_REQUIRE(pose_ok(T), "non-finite pose");
  const size_t np = pp ? pp->n : 0, nl = pl ? pl->n : 0;
  for (int i = 0; i < 36; i++) cov[i] = (i % 7 == 0) ? 1e6 : 0.0;
  if (np + nl == 0) return MH_OK;  // "no pairings -> no estimation": diag(1e6)
  MH_TRY(set_device(ctx));
  MH_TRY(ensure_state(ctx));
  hipStream_t s = ctx->stream;
  MH_HIP(mh::wait_stream(s));
  size_t sp = 0, sl = 0;
  if (np) {
    const float* arrs[3] = {pp->lx, pp->ly, pp->lz};
    MH_TRY(stage_soa(ctx, ctx->build_a, arrs, 3, np, mem, &sp));
  }
  if (nl) {
    const float* arrs[6] = {pl->lx, pl->ly, pl->lz, pl->nx, pl->ny, pl->nz};
    MH_TRY(stage_soa(ctx, ctx->build_b, arrs, 6, nl, mem, &sl));
  }
  MH_TRY(ensure_pair_buffers(ctx, np));
  const uint32_t nbp = np ? nblk(np) : 0, nbl = nl ? nblk(nl) : 0;
  MH_TRY(ctx->partials_b.reserve((size_t)kGenN * (nbl ? nbl : 1) * sizeof(double)));
  init_state(ctx->h_state, T);
  MH_HIP(hipMemcpyAsync(ctx->d_state, ctx->h_state, sizeof(IcpDeviceState), hipMemcpyHostToDevice, s));
  if (np) MH_HIP(hipMemsetAsync(ctx->pair_gidx.p, 0, np * sizeof(uint32_t), s));  // all valid
  {
    MatchK mk0{};
    SolveK sk0{};
    sk0.cov_hx = findif_xyz;
    sk0.cov_ha = findif_ang;
    MH_TRY(upload_params(ctx, mk0, sk0));
  }
  hipLaunchKernelGGL(k_cov_prepare, dim3(1), dim3(64), 0, s, ctx->d_state, &ctx->d_params->sk, 1u);
  const float* L = ctx->build_a.as<float>();
  const float* P = ctx->build_b.as<float>();
  if (np)
    hipLaunchKernelGGL(k_cov_accum, dim3(nbp), dim3(kBlock), 0, s, ctx->d_state, 1u, L, L + sp, L + 2 * sp, (uint32_t)np,
                       ctx->pair_gidx.as<uint32_t>(), ctx->partials.as<double>(), nbp);
  if (nl)
    hipLaunchKernelGGL(k_cov_accum_pl, dim3(nbl), dim3(kBlock), 0, s, ctx->d_state, P, P + 3 * sl, (uint32_t)nl,
                       (uint32_t)sl, ctx->partials_b.as<double>(), nbl);
  hipLaunchKernelGGL(k_cov_finalize, dim3(1), dim3(kSolveThreads), 0, s, ctx->d_state, 1u, ctx->partials.as<double>(), nbp, nbp,
                     ctx->partials_b.as<double>(), nbl, nbl);
  MH_HIP(hipGetLastError());
  MH_HIP(hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(IcpDeviceState), hipMemcpyDeviceToHost, s));
  MH_HIP(mh::wait_stream(s));
  for (int i = 0; i < 36; i++) cov[i] = ctx->h_state->cov[i];
  return MH_OK;
}

}  // extern "C"
