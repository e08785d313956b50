// mh_k_accum.h -- Solver_GaussNewton's sums [U] (lidar3d-default.yaml:184-190): robust weight + the 18 moment sums of the stored
// point-to-point pairings at the current pose (k_accum bodies).
#pragma once

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
