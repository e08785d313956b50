// mh_nn_device.h -- device-side building blocks of the correspondence search and of the Gauss-Newton
// accumulation.  Everything that decides WHICH point pairs (transform, voxel index, fp32 distance,
// comparison order) is written un-fused and in a fixed order so that the result is bit-identical to
// the CPU restatement of the reference algorithm (this file is compiled with -ffp-contract=off).
#pragma once
#include "mh_internal.h"

namespace mh {

constexpr uint32_t kNoMatch = 0xFFFFFFFFu;

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
  float4 pt;  // nearest map point {x,y,z,src}
  float d2;
  bool found;
};

// NearestNeighborsCapable::nn_single_search [U] on the hashed voxel map: visit the 3x3x3 voxel block
// around voxel(q) in x-outer / y-middle / z-inner order, points in insertion order, strict '<' keeps
// the first minimum (SURVEY 8a row a8, App.B U2/U3).
__device__ __forceinline__ NNResult nn_single_search(const MapView& m, float qx, float qy, float qz) {
  NNResult r;
  r.d2 = __builtin_inff();
  r.found = false;
  r.pt = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!(isfinite(qx) && isfinite(qy) && isfinite(qz))) return r;
  const float lim = 1.0e6f;
  if (!(fabsf(qx * m.inv_vs) < lim && fabsf(qy * m.inv_vs) < lim && fabsf(qz * m.inv_vs) < lim)) return r;
  const int cx = voxel_of(qx, m.inv_vs, m.trunc), cy = voxel_of(qy, m.inv_vs, m.trunc), cz = voxel_of(qz, m.inv_vs, m.trunc);
#pragma unroll 1
  for (int ix = -1; ix <= 1; ix++) {
#pragma unroll 1
    for (int iy = -1; iy <= 1; iy++) {
      // the three z-neighbours are probed together: three independent slot loads in flight
      MapSlot s[3];
      uint32_t h[3];
      unsigned long long key[3];
#pragma unroll
      for (int iz = 0; iz < 3; iz++) {
        key[iz] = pack_key(cx + ix, cy + iy, cz + iz - 1);
        h[iz] = hash_key(key[iz]) & m.mask;
        s[iz] = m.slots[h[iz]];
      }
#pragma unroll
      for (int iz = 0; iz < 3; iz++) {
        // linear probing until the key or an empty slot is met
        while (s[iz].key != key[iz] && s[iz].key != kEmptyKey) {
          h[iz] = (h[iz] + 1) & m.mask;
          s[iz] = m.slots[h[iz]];
        }
        if (s[iz].key == key[iz]) {
          const float4* __restrict__ p = m.pts + s[iz].first;
          const uint32_t cnt = s[iz].count;
          for (uint32_t j = 0; j < cnt; j++) {
            const float4 c = p[j];
            const float dx = c.x - qx, dy = c.y - qy, dz = c.z - qz;
            const float d2 = (dx * dx + dy * dy) + dz * dz;  // fp32, un-fused, this order
            if (d2 < r.d2) {
              r.d2 = d2;
              r.pt = c;
              r.found = true;
            }
          }
        }
      }
    }
  }
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
