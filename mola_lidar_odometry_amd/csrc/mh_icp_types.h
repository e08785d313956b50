// mh_icp_types.h -- what the kernels of the ICP loop share: the device-resident state and parameter blocks, the job descriptor of
// the lock-step (*_b) kernels, the fixed-shape workgroup reductions.  Included by mh_icp.hip (one translation unit: a __global__
// kernel has to be defined where it is launched).
#pragma once

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
  double hook_cos_rot;  // cos(hook_rot) for 0 < hook_rot < 3 (else NaN): the hook's rotation test is decided from the trace where that is safe
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
