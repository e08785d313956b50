// mh_k_cov.h -- mp2p_icp::covariance [U] (LidarOdometry.cpp:1009): central-difference Jacobian of the stacked residuals, (A^T A)^-1.
#pragma once

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
