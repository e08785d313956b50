// mh_nn_device.h -- device-side building blocks of the correspondence search and of the Gauss-Newton
// accumulation.  Everything that decides WHICH point pairs (transform, voxel index, fp32 distance,
// comparison order) is written un-fused and in a fixed order so that the result is bit-identical to
// the CPU restatement of the reference algorithm (this file is compiled with -ffp-contract=off).
#pragma once
#include "mh_internal.h"

namespace mh {

constexpr uint32_t kNoMatch = 0xFFFFFFFFu;

// native clang vectors: a plain dwordx4 load into registers (HIP's uint4/float4 are union structs whose
// copies become memcpy's that keep arrays of them in scratch memory)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// p' = (float)(R*l + t): double pose x float point, rounded once to float
// (Matcher_Points_Base::transform_local_to_global [U] -> CPose3D::composePoint; SURVEY App.B U4)
__device__ __forceinline__ void transform_point(const double* __restrict__ T, float lx, float ly, float lz, float& gx,
                                                float& gy, float& gz) {
  const double x = lx, y = ly, z = lz;
  gx = (float)(((T[0] * x + T[1] * y) + T[2] * z) + T[3]);
  gy = (float)(((T[4] * x + T[5] * y) + T[6] * z) + T[7]);
  gz = (float)(((T[8] * x + T[9] * y) + T[10] * z) + T[11]);
}

__device__ __forceinline__ int voxel_of(float c, float inv_vs, uint32_t trunc) {
  const float s = c * inv_vs;
  return trunc ? (int)s : (int)floorf(s);
}

struct NNResult {
  f32x4 pt;   // nearest map point {x,y,z,src}
  float d2;
  bool found;
};

// NearestNeighborsCapable::nn_single_search [U] on the hashed voxel map: visit the 3x3x3 voxel block
// around voxel(q) in x-outer / y-middle / z-inner order, points in insertion order, strict '<' keeps
// the first minimum (SURVEY 8a row a8, App.B U2/U3).
//
// Memory-level parallelism is what this kernel lives on (one scan only fills ~2 waves per SIMD, so
// the dependent-load chain per lane is the critical path):
//  * all 27 hash slots are requested before any is inspected (27 independent dwordx4 loads);
//  * point records are stored in ascending packed-key order with z in the low bits, so the records
//    of the (up to) three z-neighbours of one (x,y) column are CONTIGUOUS in HBM: the 27 voxel
//    scans collapse into 9 runs, visited in exactly the reference order;
//  * each run is read four records at a time (indices clamped to the run, so the tail re-reads the
//    last record, which a strict '<' can never select twice).
__device__ __forceinline__ NNResult nn_single_search(const MapView& m, float qx, float qy, float qz) {
  NNResult r;
  r.d2 = __builtin_inff();
  r.found = false;
  r.pt = (f32x4)(0.f);
  if (!(isfinite(qx) && isfinite(qy) && isfinite(qz))) return r;
  const float lim = 1.0e6f;
  if (!(fabsf(qx * m.inv_vs) < lim && fabsf(qy * m.inv_vs) < lim && fabsf(qz * m.inv_vs) < lim)) return r;
  const int cx = voxel_of(qx, m.inv_vs, m.trunc), cy = voxel_of(qy, m.inv_vs, m.trunc), cz = voxel_of(qz, m.inv_vs, m.trunc);
  const unsigned long long kbase = pack_key(cx - 1, cy - 1, cz - 1);
  const u32x4* __restrict__ slots4 = reinterpret_cast<const u32x4*>(m.slots);  // one dwordx4 per slot
  u32x4 s[27];
#pragma unroll
  for (int c = 0; c < 27; c++) {
    // key(cx+ix, cy+iy, cz+iz) = kbase + (ix+1)<<42 + (iy+1)<<21 + (iz+1): no carries inside the checked range
    const unsigned long long key = kbase + ((unsigned long long)(c / 9) << 42) + ((unsigned long long)((c / 3) % 3) << 21) +
                                   (unsigned long long)(c % 3);
    s[c] = slots4[hash_key(key) & m.mask];
  }
  // resolve the 27 probes into 9 (first, count) runs right away: the slots die here, only 18
  // registers stay live across the distance loops
  uint32_t first9[9], cnt9[9];
#pragma unroll
  for (int col = 0; col < 9; col++) {
    uint32_t first = 0, cnt = 0;
#pragma unroll
    for (int iz = 0; iz < 3; iz++) {
      const int c = col * 3 + iz;
      const unsigned long long key = kbase + ((unsigned long long)(c / 9) << 42) + ((unsigned long long)((c / 3) % 3) << 21) +
                                     (unsigned long long)(c % 3);
      u32x4 sl = s[c];  // {key lo, key hi, first, count}
      unsigned long long sk = ((unsigned long long)sl.y << 32) | sl.x;
      if (sk != key && sk != kEmptyKey) {  // rare: linear probing past a collision
        uint32_t h = hash_key(key) & m.mask;
        do {
          h = (h + 1) & m.mask;
          sl = slots4[h];
          sk = ((unsigned long long)sl.y << 32) | sl.x;
        } while (sk != key && sk != kEmptyKey);
      }
      if (sk == key) {
        if (cnt == 0) first = sl.z;
        cnt += sl.w;
      }
    }
    first9[col] = first;
    cnt9[col] = cnt;
  }
#pragma unroll
  for (int col = 0; col < 9; col++) {
    const uint32_t first = first9[col], cnt = cnt9[col];
    const f32x4* __restrict__ p = reinterpret_cast<const f32x4*>(m.pts) + first;
    for (uint32_t j = 0; j < cnt; j += 4) {
      const uint32_t last = cnt - 1;
      const f32x4 c0 = p[j];
      const f32x4 c1 = p[min(j + 1, last)];
      const f32x4 c2 = p[min(j + 2, last)];
      const f32x4 c3 = p[min(j + 3, last)];
      {
        const float dx = c0.x - qx, dy = c0.y - qy, dz = c0.z - qz;
        const float d2 = (dx * dx + dy * dy) + dz * dz;  // fp32, un-fused, this order
        if (d2 < r.d2) { r.d2 = d2; r.pt = c0; }
      }
      {
        const float dx = c1.x - qx, dy = c1.y - qy, dz = c1.z - qz;
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        if (d2 < r.d2) { r.d2 = d2; r.pt = c1; }
      }
      {
        const float dx = c2.x - qx, dy = c2.y - qy, dz = c2.z - qz;
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        if (d2 < r.d2) { r.d2 = d2; r.pt = c2; }
      }
      {
        const float dx = c3.x - qx, dy = c3.y - qy, dz = c3.z - qz;
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        if (d2 < r.d2) { r.d2 = d2; r.pt = c3; }
      }
    }
  }
  r.found = r.d2 < __builtin_inff();
  return r;
}

// ---- robust kernels (mp2p_icp::create_robust_kernel [U], lidar3d-default.yaml:188-190) ---------
__device__ __forceinline__ double robust_weight(uint32_t kernel, double c, double e2) {
  switch (kernel) {
    case MH_KERNEL_GM_C4: { const double c2 = c * c, d = c2 + e2; return (c2 * c2) / (d * d); }
    case MH_KERNEL_GM_KISS: { const double d = c + e2; return (c * c) / (d * d); }
    case MH_KERNEL_GM_BARRON: { const double d = e2 / (4.0 * c * c) + 1.0; return 1.0 / (d * d); }
    case MH_KERNEL_CAUCHY: { const double c2 = c * c; return c2 / (c2 + e2); }
    case MH_KERNEL_GM_C2: { const double c2 = c * c, d = c2 + e2; return c2 / (d * d); }
    default: return 1.0;
  }
}

// ---- point-to-point accumulator ---------------------------------------------------------------
// With J = [R | -R[l]x] (right perturbation T*exp(eps)) the normal equations only need, per pair,
//   r = R^T e,  w,  and the moments of l (SURVEY Appendix A closed form):
//   H_tt = (sum w) I     H_tw = -[sum w l]x     H_ww = sum w (|l|^2 I - l l^T)
//   g_t  = sum w r       g_w  = sum w (l x r)
// 18 running sums instead of 27, all fp64.
constexpr int kAccN = 18;
struct Acc {
  double v[kAccN];  // 0:sw 1-3:swl 4-9:M(xx,yy,zz,xy,xz,yz) 10-12:swr 13-15:sw(lxr) 16:cost 17:count
};

__device__ __forceinline__ void acc_zero(Acc& a) {
#pragma unroll
  for (int i = 0; i < kAccN; i++) a.v[i] = 0.0;
}

__device__ __forceinline__ void acc_pt2pt(Acc& a, const double* __restrict__ T, float lxf, float lyf, float lzf, float qxf,
                                          float qyf, float qzf, uint32_t kernel, double kparam, double wpair) {
  const double lx = lxf, ly = lyf, lz = lzf;
  const double ex = T[0] * lx + T[1] * ly + T[2] * lz + T[3] - (double)qxf;
  const double ey = T[4] * lx + T[5] * ly + T[6] * lz + T[7] - (double)qyf;
  const double ez = T[8] * lx + T[9] * ly + T[10] * lz + T[11] - (double)qzf;
  const double e2 = ex * ex + ey * ey + ez * ez;
  const double w = wpair * robust_weight(kernel, kparam, e2);
  const double rx = T[0] * ex + T[4] * ey + T[8] * ez;  // r = R^T e
  const double ry = T[1] * ex + T[5] * ey + T[9] * ez;
  const double rz = T[2] * ex + T[6] * ey + T[10] * ez;
  a.v[0] += w;
  a.v[1] += w * lx; a.v[2] += w * ly; a.v[3] += w * lz;
  a.v[4] += w * (ly * ly + lz * lz);
  a.v[5] += w * (lx * lx + lz * lz);
  a.v[6] += w * (lx * lx + ly * ly);
  a.v[7] -= w * lx * ly; a.v[8] -= w * lx * lz; a.v[9] -= w * ly * lz;
  a.v[10] += w * rx; a.v[11] += w * ry; a.v[12] += w * rz;
  a.v[13] += w * (ly * rz - lz * ry);
  a.v[14] += w * (lz * rx - lx * rz);
  a.v[15] += w * (lx * ry - ly * rx);
  a.v[16] += w * e2;
  a.v[17] += 1.0;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// Block-wide reduction of NV doubles per thread for 256-thread blocks; the result is written by the
// first NV threads to out[0..NV).  Fixed shape -> bitwise reproducible.
template <int NV>
__device__ __forceinline__ void block_reduce_store(const double* v, double* __restrict__ out, double (*lds)[NV]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const double s = wave_sum(v[i]);
    if (lane == 0) lds[wave][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double s = lds[0][threadIdx.x];
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) s += lds[w][threadIdx.x];
    out[threadIdx.x] = s;
  }
}

}  // namespace mh
