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
// ONE translation unit (a __global__ kernel is defined where it is launched), in units:
//   mh_nn_device.h, mh_nn_flat.h   the searches: quad / row / k-best (branch and bound through the caches), plan / scan
//   mh_icp_types.h                 state and parameter blocks, the lock-step job descriptor, workgroup reductions
//   mh_k_match.h, mh_k_match_rows.h, mh_k_accum.h, mh_k_solve.h, mh_k_step.h (+ mh_loop_wave.h), mh_k_cov.h, mh_k_pairs.h
//                                  kernel bodies: matchers, accumulation, Gauss-Newton step + loop tail, the small layer's loops, covariance, pairings
//   mh_k_launch.h                  their __global__ entry points (single alignment | one job per blockIdx.y)
//   this file                      AlignJob: the chain a layer takes, loop control (one-launch loop | streaming | chunks + hipGraph), mh_icp_align
//   mh_icp_batch.inl               mh_icp_align_batch (lock-step groups)
//   mh_icp_api.inl                 matcher- / solver-granular entry points (mh_nn_search*, mh_gn_solve, mh_covariance)
//   mh_dev_variants.h              (-DMH_DEV_VARIANTS only) the tile / wave / sorted-scan matchers that lost to the product kernels
//
// Kernel sequence per ICP iteration of a LARGE layer (everything stays in HBM/L2; the host never sees pairings):
//   k_match_flat     a wave per 64 scan points: transform, exact bounded search (plan / scan), threshold test, store pairing
//   k_accum          robust weight + 18 fp64 moment sums of the stored pairings at the current pose -> partials
//   k_solve          one workgroup: ordered sum of partials, prior factor, LDL^T, T <- T(+)exp(delta),
//                    inner/outer loop bookkeeping, stall + hook tests, termination flag, next threshold
//   k_accum, k_solve (inner Gauss-Newton steps >= 1 on the SAME pairings)
// Smaller layers take shorter chains (chosen by size in AlignJob::start / enqueue_chunk):
//   <= 32 k points   k_match16: a DPP row (16 lanes) per point; up to 12 k points it also accumulates the first step
//   <= 8 k points    k_step16 x max_inner: search + sums per launch, the Gauss-Newton step carried into the next launch
//   <= 2560 points   k_icp16: the whole loop in ONE launch (single alignments); lock-step batches of point layers up to 4096
//                    points: k_icpw_b (the same loop with the plan / scan search) -- lidar3d-default.yaml's ICP layer
//   NDT maps         Matcher_Point2Plane rides in the row kernels (k_step16<true>, k_icp16<true>, k_match16<true,.>), its rows are
//                    summed alongside (k_accum_both); above 32 k points k_match_pl (one lane per point)
// No fp atomics anywhere: reductions are fixed-shape trees, so results are bitwise reproducible.
#include <math.h>
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

#include "mh_icp_types.h"
#include "mh_k_match.h"
#include "mh_k_accum.h"
#include "mh_k_match_rows.h"
#include "mh_k_solve.h"
#include "mh_k_step.h"
#include "mh_k_cov.h"
#include "mh_k_launch.h"
#include "mh_k_pairs.h"

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
mh_status upload_state_and_params(mh_ctx* ctx, const MatchK& mk, const SolveK& sk, size_t inline_sched = 0) {
  ctx->h_params->mk = mk;
  ctx->h_params->sk = sk;
  const size_t bytes = inline_sched ? kInlineSchedOffset + inline_sched * sizeof(double) : kParamsOffset + sizeof(IcpDeviceParams);
  MH_HIP(hipMemcpyAsync(ctx->d_state, ctx->h_state, bytes, hipMemcpyHostToDevice, ctx->stream));
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
    sk.hook_cos_rot = (p->hook_min_rot > 0.0 && p->hook_min_rot < 3.0) ? cos(p->hook_min_rot) : __builtin_nan("");
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
    if (streaming && use_step_chain() && !forbid_loop16 && getenv("MH_NO_LOOP16") == nullptr && !loop_holdoff(ctx->device, 0)) {
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
    if (defer_upload) {
      ctx->h_params->mk = mk;
      ctx->h_params->sk = sk;
    } else {
      MH_TRY(upload_state_and_params(ctx, mk, sk, inline_sched));
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
  // arm > 0: a loop was abandoned; arm < 0: a loop ran to its end; 0: true while the hold-off lasts (counts down).  The hold-off
  // doubles with every abandonment in a row (200, 400, ... 12 800 alignments): on a device shared with other processes a loop's
  // workgroups may never run together, and every attempt costs its 20 ms limit (MH_NO_LOOP16=1 is the switch for such
  // a deployment).
  static bool loop_holdoff(int device, int arm) {
    static std::atomic<int> left[64], streak[64];
    std::atomic<int>& v = left[(unsigned)device % 64u];
    std::atomic<int>& st = streak[(unsigned)device % 64u];
    if (arm > 0) {
      const int k = st.fetch_add(1);
      v.store(200 << (k < 6 ? k : 6));
      return true;
    }
    if (arm < 0) {
      st.store(0);
      return false;
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
    hipStream_t s = ctx->stream;
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
    MH_TRY(enqueue_tail(/*cov_prepared=*/true));
    const hipError_t we = mh::wait_event(ctx->ev_poll);
    loop_release();
    MH_HIP(we);
    const IcpDeviceState* h = ctx->h_state;
    if (!h->done || h->handover_timeouts) {
      g_loop16_fallbacks.fetch_add(1);
      if (getenv("MH_LOOP16_TEST_ABANDON") == nullptr) loop_holdoff(ctx->device, 1);
      return MH_OK;  // (not finished)
    }
    loop_holdoff(ctx->device, -1);
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

  mh_status enqueue_tail(bool cov_prepared = false) {  // cov_prepared: k_icp16 has done k_cov_prepare's part
    MH_TRY(set_device(ctx));
    hipStream_t s = ctx->stream;
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

#include "mh_icp_batch.inl"  // mh_icp_align_batch

#include "mh_icp_api.inl"    // mh_nn_search*, mh_gn_solve, mh_covariance

}  // extern "C"
