// mh_k_match.h -- Matcher_Points_DistanceThreshold [U] (lidar3d-default.yaml:195-204): the one-lane matcher of the matcher-granular
// entry points (k_match), pairingsPerPoint > 1 (k_match_kbest), Matcher_Point2Plane on a plain point layer (k_match_pl_knn), the
// debug build's phase stamps, and the quad matcher's kernel body (k_match4_body; the search itself: mh_nn_device.h).
#pragma once

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
