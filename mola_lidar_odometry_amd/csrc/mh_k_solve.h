// mh_k_solve.h -- the Gauss-Newton step and the tail of ICP::align's loop body [U]: both kinds of rows in one launch (k_accum_both),
// the ordered sums of the partials (reduce_rows), solve_body (prior factor, 6x6 LDL^T, SE(3) retraction, inner / outer loop
// bookkeeping, stall and hook tests, termination, next threshold) and k_solve's body.
#pragma once

// ================================================================================================
// k_solve: one wave.  Ordered reduction of the block partials, prior factor, LDL^T solve, SE(3)
// retraction, inner/outer loop bookkeeping (optimal_tf_gauss_newton + the tail of ICP::align's loop).
// ================================================================================================
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
  // (sqrt(cost) <= max_cost without the square root where its square decides: max_cost is 0 in both shipped pipelines, and
  //  sqrt(c) <= 0 <=> c <= 0)
  bool target_met;
  if (k.max_cost == 0.0) target_met = cost <= 0.0;
  else if (cost > k.max_cost * k.max_cost * (1.0 + 1e-9)) target_met = false;
  else target_met = sqrt(cost) <= k.max_cost;
  if (target_met) {
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
    {  // sqrt(dn) < min_delta, decided on the squares outside a margin far above their rounding
      const double md2 = k.min_delta * k.min_delta;
      if (dn < md2 * (1.0 - 1e-9)) inner_done = true;
      else if (!(dn > md2 * (1.0 + 1e-9)) && sqrt(dn) < k.min_delta) inner_done = true;
    }
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
    // The two norms cost a lane's microsecond (compose, atan2, three square roots) and almost never sit near their limits: the
    // translation is decided on its square, the rotation angle on the trace (theta > r <=> cos theta < cos r on [0, pi]), each
    // with a margin far above the rounding of either side; only inside the margins the norms themselves are formed.  Same
    // decisions, same results (as for the stall test above).
    double tq[3], dg[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      tq[i] = Ci.R(i, 0) * Tc.t(0) + Ci.R(i, 1) * Tc.t(1) + Ci.R(i, 2) * Tc.t(2) + Ci.t(i);   // compose()'s translation, its order
      dg[i] = Ci.R(i, 0) * Tc.R(0, i) + Ci.R(i, 1) * Tc.R(1, i) + Ci.R(i, 2) * Tc.R(2, i);     // ... and the diagonal of its rotation
    }
    const double t2 = tq[0] * tq[0] + tq[1] * tq[1] + tq[2] * tq[2];
    const double lim2 = k.hook_trans * k.hook_trans;
    const double cth = 0.5 * (dg[0] + dg[1] + dg[2] - 1.0);
    int verdict = -1;  // 1: stop, 0: go on, -1: undecided
    if (t2 > lim2 * (1.0 + 1e-9) + 1e-300) verdict = 1;
    else if (t2 < lim2 * (1.0 - 1e-9) && cth > k.hook_cos_rot + 1e-9) verdict = 0;
    else if (t2 < lim2 * (1.0 - 1e-9) && cth < k.hook_cos_rot - 1e-9) verdict = 1;
    if (verdict < 0) {
      const Pose S = compose(Ci, Tc);
      double w[3];
      so3_log(S, w);
      const double ht = sqrt(S.t(0) * S.t(0) + S.t(1) * S.t(1) + S.t(2) * S.t(2));
      const double hr = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
      verdict = (ht > k.hook_trans || hr > k.hook_rot) ? 1 : 0;
    }
    if (verdict) {
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
