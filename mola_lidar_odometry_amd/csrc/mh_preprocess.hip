// mh_preprocess.hip -- scan pre-processing on the device: the observation filter chain of
// lidar3d-default.yaml:270-350 that turns the raw sensor cloud into the layers `decimated_for_map` and
// `decimated_for_icp`, so that the per-scan path is device resident from the raw points on and the re-de-skew
// inside the ICP loop (LidarOdometry.cpp:992-999) needs no host round trip.
//
//   FilterAdjustTimestamps  -> min/max reduction, applied while compacting
//   FilterDecimateVoxels    -> FirstPoint: hash insert with atomicMin(first index) per voxel, then "am I the first"
//   FilterByRange / FilterBoundingBox -> per-point predicates fused into the same flag pass
//   (flags) -> exclusive scan -> order-preserving compaction (survivors keep the raw order)
//   FilterDeskew            -> p' = Exp_SO3(w t_i) p + v t_i, fp64, rounded to float
// All of it is HBM-bound byte/index work: coalesced SoA streams, one pass per stage, atomics only on the
// (L2-resident) decimation table.
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "mh_internal.h"
#include "mh_nn_device.h"

using namespace mh;

namespace {

struct StageParams {
  float inv_res;       // 1/voxel_filter_resolution, 0 = no decimation in this stage
  uint32_t trunc;      // MH_INDEX_TRUNC
  uint32_t decimate;   // 0 when the input is smaller than minimum_input_points_to_filter
  uint32_t range_on;
  float sq_min, sq_max, cx, cy, cz;
  int32_t bbox_mode;
  float bmin[3], bmax[3];
  int32_t ts_method;   // applied to t while compacting (stage 1 only)
  float ts_offset;
  uint32_t method;     // MH_DECIMATE_*: which point of a voxel survives
};

__device__ __forceinline__ uint32_t f2ord(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
  u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
  return __uint_as_float(u);
}

// counters[0] = min(t), counters[1] = max(t) as order-preserving uints.  Grid-stride over a fixed, small grid: one
// pair of atomics per workgroup (one per wave serialised ~2 k same-address atomics: 44 us for a 120 k-point scan)
__global__ __launch_bounds__(256) void k_pp_tminmax(const float* __restrict__ t, uint32_t n, uint32_t* __restrict__ counters) {
  __shared__ uint32_t smn[4], smx[4];
  uint32_t mn = 0xFFFFFFFFu, mx = 0u;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t o = f2ord(t[i]);
    mn = min(mn, o);
    mx = max(mx, o);
  }
  for (int off = 32; off > 0; off >>= 1) {
    mn = min(mn, (uint32_t)__shfl_xor((int)mn, off));
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
  }
  if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    mn = min(min(smn[0], smn[1]), min(smn[2], smn[3]));
    mx = max(max(smx[0], smx[1]), max(smx[2], smx[3]));
    if (mn != 0xFFFFFFFFu) {
      atomicMin(&counters[0], mn);
      atomicMax(&counters[1], mx);
    }
  }
}

__device__ __forceinline__ bool pp_key(float px, float py, float pz, float inv, uint32_t trunc, unsigned long long& key,
                                       uint32_t* __restrict__ counters) {
  if (!(isfinite(px) && isfinite(py) && isfinite(pz))) return false;
  const float sx = px * inv, sy = py * inv, sz = pz * inv;
  if (!(fabsf(sx) < 1.0e6f && fabsf(sy) < 1.0e6f && fabsf(sz) < 1.0e6f)) {
    atomicOr(&counters[2], 1u);  // voxel index does not fit the 21-bit key fields
    return false;
  }
  key = pack_key(voxel_of(px, inv, trunc), voxel_of(py, inv, trunc), voxel_of(pz, inv, trunc));
  return true;
}

// FirstPoint decimation, pass 1: every point claims its voxel's table entry and records the smallest point index
__global__ void k_pp_insert(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                            uint32_t n, float inv, uint32_t trunc, unsigned long long* __restrict__ keys,
                            uint32_t* __restrict__ first_idx, uint32_t mask, uint32_t* __restrict__ counters) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long key;
  if (!pp_key(x[i], y[i], z[i], inv, trunc, key, counters)) return;
  uint32_t h = hash_key(key) & mask;
  for (;;) {
    const unsigned long long old = atomicCAS(&keys[h], kEmptyKey, key);
    if (old == kEmptyKey || old == key) break;
    h = (h + 1) & mask;
  }
  atomicMin(&first_idx[h], i);
}

// pass 2: flag = first point of its voxel (when decimating) && FilterByRange && FilterBoundingBox
__global__ void k_pp_flag(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                          uint32_t n, StageParams sp, const unsigned long long* __restrict__ keys,
                          const uint32_t* __restrict__ first_idx, uint32_t mask, uint32_t* __restrict__ counters,
                          uint32_t* __restrict__ flag, uint32_t count_slot) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < n;
  const float px = valid ? x[i] : 0.f, py = valid ? y[i] : 0.f, pz = valid ? z[i] : 0.f;
  bool keep = valid && isfinite(px) && isfinite(py) && isfinite(pz);
  if (keep && sp.decimate) {
    unsigned long long key;
    keep = pp_key(px, py, pz, sp.inv_res, sp.trunc, key, counters);
    if (keep) {
      uint32_t h = hash_key(key) & mask;
      while (keys[h] != key) h = (h + 1) & mask;  // inserted by pass 1
      keep = first_idx[h] == i;
    }
  }
  if (keep && sp.range_on) {
    const float dx = px - sp.cx, dy = py - sp.cy, dz = pz - sp.cz;
    const float sq = (dx * dx + dy * dy) + dz * dz;
    keep = sq >= sp.sq_min && sq <= sp.sq_max;
  }
  if (keep && sp.bbox_mode) {
    const bool inside = px >= sp.bmin[0] && px <= sp.bmax[0] && py >= sp.bmin[1] && py <= sp.bmax[1] &&
                        pz >= sp.bmin[2] && pz <= sp.bmax[2];
    keep = inside == (sp.bbox_mode == MH_BBOX_KEEP_INSIDE);
  }
  if (valid) flag[i] = keep ? 1u : 0u;
  const unsigned long long kept = __ballot(keep);  // survivor count: one atomic per wavefront
  if ((threadIdx.x & 63) == 0 && kept) atomicAdd(&counters[count_slot], (uint32_t)__popcll(kept));
}

__global__ void k_pp_compact(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                             const float* __restrict__ t, const uint32_t* __restrict__ src, uint32_t n,
                             const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pos, int32_t ts_method,
                             float ts_offset, const uint32_t* __restrict__ counters, float* __restrict__ ox,
                             float* __restrict__ oy, float* __restrict__ oz, float* __restrict__ ot,
                             uint32_t* __restrict__ osrc) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !flag[i]) return;
  const uint32_t o = pos[i];
  ox[o] = x[i];
  oy[o] = y[i];
  oz[o] = z[i];
  if (ot) {
    float tv = t[i];
    if (ts_method != MH_TS_NONE) {  // FilterAdjustTimestamps over ALL raw points (it runs before the decimation)
      const float tmin = ord2f(counters[0]), tmax = ord2f(counters[1]);
      const float dt = ts_method == MH_TS_MIDDLE_IS_ZERO ? 0.5f * (tmin + tmax) : tmin;
      tv = (tv - dt) + ts_offset;
    }
    ot[o] = tv;
  }
  osrc[o] = src ? src[i] : i;
}

// The same for a layer one workgroup can walk (what the sensor-range estimate reads every scan: the ~1 k-point ICP
// layer): no counters to initialise, no atomics, and the seven words go straight to page-locked HOST memory -- one
// launch and one wait instead of an upload, a launch, a download and a wait.
__global__ __launch_bounds__(1024) void k_pp_bbox_one(const float* __restrict__ x, const float* __restrict__ y,
                                                      const float* __restrict__ z, uint32_t n, uint32_t* __restrict__ host_out) {
  __shared__ uint32_t sh[16][7];
  uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u}, cnt = 0;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const float p[3] = {x[i], y[i], z[i]};
    if (isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2])) {
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const uint32_t o = f2ord(p[a]);
        mn[a] = min(mn[a], o);
        mx[a] = max(mx[a], o);
      }
      cnt++;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      mn[a] = min(mn[a], (uint32_t)__shfl_xor((int)mn[a], off));
      mx[a] = max(mx[a], (uint32_t)__shfl_xor((int)mx[a], off));
    }
    cnt += (uint32_t)__shfl_xor((int)cnt, off);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) { sh[w][a] = mn[a]; sh[w][3 + a] = mx[a]; }
    sh[w][6] = cnt;
  }
  __syncthreads();
  if (threadIdx.x < 7) {
    const int a = threadIdx.x;
    uint32_t v = sh[0][a];
    for (int q = 1; q < (int)(blockDim.x >> 6); q++) v = a < 3 ? min(v, sh[q][a]) : (a < 6 ? max(v, sh[q][a]) : v + sh[q][a]);
    host_out[a] = v;
  }
}

// counters[0..2] = min, [3..5] = max (ordered uints), [6] = finite count; grid-stride, seven atomics per workgroup
__global__ __launch_bounds__(256) void k_pp_bbox(const float* __restrict__ x, const float* __restrict__ y,
                                                 const float* __restrict__ z, uint32_t n, uint32_t* __restrict__ counters) {
  __shared__ uint32_t sh[4][7];
  uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u}, cnt = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float p[3] = {x[i], y[i], z[i]};
    if (isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2])) {
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const uint32_t o = f2ord(p[a]);
        mn[a] = min(mn[a], o);
        mx[a] = max(mx[a], o);
      }
      cnt++;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      mn[a] = min(mn[a], (uint32_t)__shfl_xor((int)mn[a], off));
      mx[a] = max(mx[a], (uint32_t)__shfl_xor((int)mx[a], off));
    }
    cnt += (uint32_t)__shfl_xor((int)cnt, off);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) { sh[w][a] = mn[a]; sh[w][3 + a] = mx[a]; }
    sh[w][6] = cnt;
  }
  __syncthreads();
  if (threadIdx.x < 7) {
    const int a = threadIdx.x;
    uint32_t v = sh[0][a];
    for (int q = 1; q < 4; q++) v = a < 3 ? min(v, sh[q][a]) : (a < 6 ? max(v, sh[q][a]) : v + sh[q][a]);
    const uint32_t total = sh[0][6] + sh[1][6] + sh[2][6] + sh[3][6];
    if (total) {
      if (a < 3) atomicMin(&counters[a], v);
      else if (a < 6) atomicMax(&counters[a], v);
      else atomicAdd(&counters[6], v);
    }
  }
}

struct Twist { double v[6]; };

__device__ __forceinline__ void deskew_point(const Twist& tw, float x, float y, float z, float tf, float& gx, float& gy, float& gz) {
  const double dt = (double)tf;
  const double xi[6] = {0.0, 0.0, 0.0, tw.v[3] * dt, tw.v[4] * dt, tw.v[5] * dt};
  Pose p = se3_exp(xi);  // zero translation part: pure Exp_SO3(w dt)
  p.t(0) = tw.v[0] * dt;
  p.t(1) = tw.v[1] * dt;
  p.t(2) = tw.v[2] * dt;
  transform_point(p.m, x, y, z, gx, gy, gz);
}

__global__ void k_pp_deskew(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                            const float* __restrict__ t, const uint32_t* __restrict__ src, uint32_t n, Twist tw,
                            float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz,
                            float* __restrict__ ot, uint32_t* __restrict__ osrc) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float tf = t[i];
  ot[i] = tf;
  if (src) osrc[i] = src[i];
  float gx, gy, gz;
  deskew_point(tw, x[i], y[i], z[i], tf, gx, gy, gz);
  ox[i] = gx;
  oy[i] = gy;
  oz[i] = gz;
}

// Both layers of a scan in one launch, and the bounding box of the de-skewed SMALL layer (what the sensor-range estimate
// reads next, LidarOdometry.cpp:744) straight into page-locked host memory: workgroup 0 walks the small layer and
// reduces its box, the others take 1024 points of the large layer each.  One launch and one wait instead of three
// launches and a wait -- on a path where every launch is a dependent step of the per-scan chain.
struct DeskewLayer {
  const float *x, *y, *z, *t;
  const uint32_t* src;
  float *ox, *oy, *oz, *ot;
  uint32_t* osrc;
  uint32_t n;
};

__global__ __launch_bounds__(1024) void k_pp_deskew_pair(DeskewLayer big, DeskewLayer small, Twist tw, uint32_t* __restrict__ host_out) {
  if (blockIdx.x > 0) {
    const uint32_t i = (blockIdx.x - 1) * 1024 + threadIdx.x;
    if (i >= big.n) return;
    const float tf = big.t[i];
    big.ot[i] = tf;
    if (big.src) big.osrc[i] = big.src[i];
    float gx, gy, gz;
    deskew_point(tw, big.x[i], big.y[i], big.z[i], tf, gx, gy, gz);
    big.ox[i] = gx;
    big.oy[i] = gy;
    big.oz[i] = gz;
    return;
  }
  __shared__ uint32_t sh[16][7];
  uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u}, cnt = 0;
  for (uint32_t i = threadIdx.x; i < small.n; i += 1024) {
    const float tf = small.t[i];
    small.ot[i] = tf;
    if (small.src) small.osrc[i] = small.src[i];
    float p[3];
    deskew_point(tw, small.x[i], small.y[i], small.z[i], tf, p[0], p[1], p[2]);
    small.ox[i] = p[0];
    small.oy[i] = p[1];
    small.oz[i] = p[2];
    if (isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2])) {
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const uint32_t o = f2ord(p[a]);
        mn[a] = min(mn[a], o);
        mx[a] = max(mx[a], o);
      }
      cnt++;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      mn[a] = min(mn[a], (uint32_t)__shfl_xor((int)mn[a], off));
      mx[a] = max(mx[a], (uint32_t)__shfl_xor((int)mx[a], off));
    }
    cnt += (uint32_t)__shfl_xor((int)cnt, off);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) { sh[w][a] = mn[a]; sh[w][3 + a] = mx[a]; }
    sh[w][6] = cnt;
  }
  __syncthreads();
  if (threadIdx.x < 7) {
    const int a = threadIdx.x;
    uint32_t v = sh[0][a];
    for (int q = 1; q < 16; q++) v = a < 3 ? min(v, sh[q][a]) : (a < 6 ? max(v, sh[q][a]) : v + sh[q][a]);
    host_out[a] = v;
  }
}

// interleaved records (step/offsets in 4-byte words) -> SoA
__global__ void k_pp_deinterleave(const uint32_t* __restrict__ data, uint32_t n, uint32_t step, uint32_t ox, uint32_t oy,
                                  uint32_t oz, int32_t ot, float* __restrict__ x, float* __restrict__ y,
                                  float* __restrict__ z, float* __restrict__ t) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* rec = data + (size_t)i * step;
  x[i] = __uint_as_float(rec[ox]);
  y[i] = __uint_as_float(rec[oy]);
  z[i] = __uint_as_float(rec[oz]);
  if (ot >= 0) t[i] = __uint_as_float(rec[ot]);
}

inline uint32_t nblk(size_t n, uint32_t b) { return (uint32_t)((n + b - 1) / b); }

// ---- the filter chain over several scans at once (one launch per step, blockIdx.y = scan) -------------------------
// Everything the host used to learn between the steps (how many points stage 1 left, whether stage 2 decimates) stays
// on the device: the outputs are allocated for the raw size, stage 2 runs over that many threads with the live count
// read from the counters, and ONE read-back at the end of the chain brings the counts of all scans.
struct PpJob {
  const float *x, *y, *z, *t;  // raw input
  const uint32_t* src;
  uint32_t n;                  // raw count (also the capacity of the outputs)
  uint32_t off, cap;           // this scan's range in the shared flag / position arrays (cap = n rounded up to 64)
  StageParams s1, s2;
  uint32_t min_points;         // stage 2 decimates only when stage 1 left at least this many (minimum_input_points_to_filter)
  uint32_t want_icp, want_t;
  unsigned long long *keys1, *keys2;
  uint32_t *first1, *first2;
  uint32_t tsize1, tsize2;     // table sizes (powers of two; 0 = the stage never decimates)
  uint32_t* counters;          // 8 words: tmin, tmax, range flag, survivors of stage 1, of stage 2
  float *mx, *my, *mz, *mt;    // out_map
  uint32_t* msrc;
  float *ix, *iy, *iz, *it;    // out_icp
  uint32_t* isrc;
};

__global__ __launch_bounds__(256) void k_pp_init_b(const PpJob* __restrict__ jobs) {
  const PpJob& j = jobs[blockIdx.y];
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  // (pointers that arrive through the descriptor are generic to the compiler: G() spells the global space out, or every
  // access below becomes a flat_ instruction -- the decimation's CAS loop ran at half its speed that way)
  if (tid < 8) G(j.counters)[tid] = tid == 0 ? 0xFFFFFFFFu : 0u;
  unsigned long long MH_AS_GLOBAL* k1 = G(j.keys1);
  unsigned long long MH_AS_GLOBAL* k2 = G(j.keys2);
  uint32_t MH_AS_GLOBAL* f1 = G(j.first1);
  uint32_t MH_AS_GLOBAL* f2 = G(j.first2);
  for (uint32_t i = tid; i < j.tsize1; i += stride) {
    k1[i] = kEmptyKey;
    f1[i] = 0xFFFFFFFFu;
  }
  for (uint32_t i = tid; i < j.tsize2; i += stride) {
    k2[i] = kEmptyKey;
    f2[i] = 0xFFFFFFFFu;
  }
}

__global__ __launch_bounds__(256) void k_pp_tminmax_b(const PpJob* __restrict__ jobs) {
  const PpJob& j = jobs[blockIdx.y];
  if (!j.t || j.s1.ts_method == MH_TS_NONE) return;
  __shared__ uint32_t smn[4], smx[4];
  uint32_t mn = 0xFFFFFFFFu, mx = 0u;
  const float MH_AS_GLOBAL* gt = G(j.t);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < j.n; i += gridDim.x * blockDim.x) {
    const uint32_t o = f2ord(gt[i]);
    mn = min(mn, o);
    mx = max(mx, o);
  }
  for (int off = 32; off > 0; off >>= 1) {
    mn = min(mn, (uint32_t)__shfl_xor((int)mn, off));
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
  }
  if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    mn = min(min(smn[0], smn[1]), min(smn[2], smn[3]));
    mx = max(max(smx[0], smx[1]), max(smx[2], smx[3]));
    if (mn != 0xFFFFFFFFu) {
      atomicMin(&j.counters[0], mn);
      atomicMax(&j.counters[1], mx);
    }
  }
}

// what a stage reads: stage 1 the raw scan, stage 2 what stage 1 wrote (its count lives in the counters)
template <int STAGE>
__device__ __forceinline__ void pp_stage_view(const PpJob& j, const float*& x, const float*& y, const float*& z, uint32_t& n,
                                              StageParams& sp, unsigned long long*& keys, uint32_t*& first, uint32_t& mask) {
  if (STAGE == 1) {
    x = j.x; y = j.y; z = j.z;
    n = j.n;
    sp = j.s1;
    keys = j.keys1; first = j.first1; mask = j.tsize1 - 1;
  } else {
    x = j.mx; y = j.my; z = j.mz;
    n = G(j.counters)[3];
    sp = j.s2;
    if (n < j.min_points) sp.decimate = 0;
    keys = j.keys2; first = j.first2; mask = j.tsize2 - 1;
  }
}

template <int STAGE>
__global__ __launch_bounds__(256) void k_pp_insert_b(const PpJob* __restrict__ jobs) {
  const PpJob& j = jobs[blockIdx.y];
  if (STAGE == 2 && !j.want_icp) return;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x * blockDim.x >= j.n) return;  // (whole workgroup beyond the scan: the lanes below stay together)
  const float *x, *y, *z;
  uint32_t n, mask, *first;
  unsigned long long* keys;
  StageParams sp;
  pp_stage_view<STAGE>(j, x, y, z, n, sp, keys, first, mask);
  if (!sp.decimate) return;  // (uniform)
  unsigned long long key = kEmptyKey;
  bool live = i < n;
  if (live) live = pp_key(G(x)[i], G(y)[i], G(z)[i], sp.inv_res, sp.trunc, key, j.counters);
  if (!live) key = kEmptyKey;
  // A sensor delivers its points ring by ring, so neighbours in the array are neighbours in space: of a run of lanes
  // with the same voxel only the first touches the table -- it holds the run's smallest index, which is all the others
  // could contribute (atomicMin), and the claim of the entry is the same.  Fewer same-address atomics, same table.
  const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
  const uint32_t plo = (uint32_t)__shfl_up((int)lo, 1), phi = (uint32_t)__shfl_up((int)hi, 1);
  const bool same_as_previous_lane = (threadIdx.x & 63) != 0 && plo == lo && phi == hi;
  if (!live || same_as_previous_lane) return;
  unsigned long long MH_AS_GLOBAL* gkeys = G(keys);
  uint32_t h = hash_key(key) & mask;
  for (;;) {
    unsigned long long old = kEmptyKey;
    if (__hip_atomic_compare_exchange_strong(&gkeys[h], &old, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || old == key)
      break;
    h = (h + 1) & mask;
  }
  (void)__hip_atomic_fetch_min(&G(first)[h], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int STAGE>
__global__ __launch_bounds__(256) void k_pp_flag_b(const PpJob* __restrict__ jobs, uint32_t* __restrict__ flag) {
  const PpJob& j = jobs[blockIdx.y];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x * blockDim.x >= j.cap) return;  // (whole workgroup beyond this scan's range)
  if (STAGE == 2 && !j.want_icp) {
    if (i < j.cap) flag[j.off + i] = 0u;
    return;
  }
  const float *x, *y, *z;
  uint32_t n, mask, *first;
  unsigned long long* keys;
  StageParams sp;
  pp_stage_view<STAGE>(j, x, y, z, n, sp, keys, first, mask);
  const bool valid = i < n;
  const float px = valid ? G(x)[i] : 0.f, py = valid ? G(y)[i] : 0.f, pz = valid ? G(z)[i] : 0.f;
  bool keep = valid && isfinite(px) && isfinite(py) && isfinite(pz);
  if (keep && sp.decimate) {
    unsigned long long key;
    keep = pp_key(px, py, pz, sp.inv_res, sp.trunc, key, j.counters);
    if (keep) {
      const unsigned long long MH_AS_GLOBAL* gkeys = G(keys);
      uint32_t h = hash_key(key) & mask;
      while (gkeys[h] != key) h = (h + 1) & mask;  // inserted by k_pp_insert_b
      keep = G(first)[h] == i;
    }
  }
  if (keep && sp.range_on) {
    const float dx = px - sp.cx, dy = py - sp.cy, dz = pz - sp.cz;
    const float sq = (dx * dx + dy * dy) + dz * dz;
    keep = sq >= sp.sq_min && sq <= sp.sq_max;
  }
  if (keep && sp.bbox_mode) {
    const bool inside = px >= sp.bmin[0] && px <= sp.bmax[0] && py >= sp.bmin[1] && py <= sp.bmax[1] &&
                        pz >= sp.bmin[2] && pz <= sp.bmax[2];
    keep = inside == (sp.bbox_mode == MH_BBOX_KEEP_INSIDE);
  }
  if (i < j.cap) flag[j.off + i] = keep ? 1u : 0u;  // (zeros up to the end of the range: the scan runs over all of it)
  // (the survivor count is the scan's total: k_pp_compact_b writes it -- an atomic per wavefront on ONE address was most
  // of this kernel's time, 2 k of them for a 120 k-point scan)
}

// DecimateMethod::ClosestToAverage [U]: the survivor of a voxel is the point closest to the voxel's mean, the mean being the
// float sum of its points IN INPUT ORDER times 1.0f / count -- an order no atomic gives.  So for a stage with this method the
// points of ONE scan are sorted by voxel key (stable radix sort: input order inside a voxel), the first lane of every run of
// equal keys walks its run twice (sum, then closest; strictly closer wins, i.e. the first of equals) and writes the winner's
// index where k_pp_insert_b left the voxel's smallest one: k_pp_flag_b keeps "the index the table names" either way.
constexpr unsigned long long kCtaInvalid = ~0ull;
template <int STAGE>
__global__ __launch_bounds__(256) void k_cta_keys(const PpJob* __restrict__ jobs, uint32_t job, unsigned long long* __restrict__ keys,
                                                  uint32_t* __restrict__ idx) {
  const PpJob& j = jobs[job];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= j.n) return;
  const float *x, *y, *z;
  uint32_t n, mask, *first;
  unsigned long long* tk;
  StageParams sp;
  pp_stage_view<STAGE>(j, x, y, z, n, sp, tk, first, mask);
  unsigned long long key = kCtaInvalid;
  if (sp.decimate && i < n) {
    unsigned long long k;
    if (pp_key(G(x)[i], G(y)[i], G(z)[i], sp.inv_res, sp.trunc, k, j.counters)) key = k;
  }
  keys[i] = key;
  idx[i] = i;
}
template <int STAGE>
__global__ __launch_bounds__(256) void k_cta_choose(const PpJob* __restrict__ jobs, uint32_t job,
                                                    const unsigned long long* __restrict__ keys_s, const uint32_t* __restrict__ perm) {
  const PpJob& j = jobs[job];
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= j.n) return;
  const unsigned long long key = keys_s[p];
  if (key == kCtaInvalid || (p > 0 && keys_s[p - 1] == key)) return;  // not the head of a voxel's run
  const float *x, *y, *z;
  uint32_t n, mask, *first;
  unsigned long long* tk;
  StageParams sp;
  pp_stage_view<STAGE>(j, x, y, z, n, sp, tk, first, mask);
  const float MH_AS_GLOBAL* gx = G(x);
  const float MH_AS_GLOBAL* gy = G(y);
  const float MH_AS_GLOBAL* gz = G(z);
  // (the sum is a serial chain by definition -- a voxel next to the sensor holds thousands of points -- but its operands are not:
  //  eight points' indices and coordinates are requested together, then added one after the other)
  uint32_t cnt;
  {  // the end of the run of equal keys: gallop, then bisect (sorted keys)
    uint32_t step = 1;
    while (p + step < j.n && keys_s[p + step] == key) step *= 2u;
    uint32_t lo = p + step / 2u, hi = p + step < j.n ? p + step : j.n;  // keys_s[lo] == key; hi: first position known to differ, or the end
    while (hi - lo > 1u) {
      const uint32_t mid = lo + (hi - lo) / 2u;
      if (keys_s[mid] == key) lo = mid;
      else hi = mid;
    }
    cnt = hi - p;
  }
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (uint32_t q = 0; q < cnt; q += 8u) {
    uint32_t ii[8];
    float vx[8], vy[8], vz[8];
#pragma unroll
    for (int u = 0; u < 8; u++) ii[u] = perm[p + (q + u < cnt ? q + u : cnt - 1u)];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      vx[u] = gx[ii[u]];
      vy[u] = gy[ii[u]];
      vz[u] = gz[ii[u]];
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (q + u < cnt) {
        sx += vx[u];
        sy += vy[u];
        sz += vz[u];
      }
  }
  const float inv_n = 1.0f / (float)cnt;
  const float mx = sx * inv_n, my = sy * inv_n, mz = sz * inv_n;
  float best = 0.f;
  uint32_t best_i = perm[p];
  for (uint32_t q = 0; q < cnt; q += 8u) {
    uint32_t ii[8];
    float vx[8], vy[8], vz[8];
#pragma unroll
    for (int u = 0; u < 8; u++) ii[u] = perm[p + (q + u < cnt ? q + u : cnt - 1u)];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      vx[u] = gx[ii[u]];
      vy[u] = gy[ii[u]];
      vz[u] = gz[ii[u]];
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (q + u < cnt) {
        const float dx = vx[u] - mx, dy = vy[u] - my, dz = vz[u] - mz;
        const float e = (dx * dx + dy * dy) + dz * dz;
        if (q + u == 0 || e < best) {
          best = e;
          best_i = ii[u];
        }
      }
  }
  const unsigned long long MH_AS_GLOBAL* gkeys = G(tk);
  uint32_t h = hash_key(key) & mask;
  while (gkeys[h] != key) h = (h + 1) & mask;  // inserted by k_pp_insert_b
  G(first)[h] = best_i;
}

// pos = exclusive scan of the flags of ALL scans in a row: a survivor's place in its own output is pos - pos[first of scan]
template <int STAGE>
__global__ __launch_bounds__(256) void k_pp_compact_b(const PpJob* __restrict__ jobs, const uint32_t* __restrict__ flag,
                                                      const uint32_t* __restrict__ pos) {
  const PpJob& j = jobs[blockIdx.y];
  if (STAGE == 2 && !j.want_icp) return;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && j.cap)  // survivors of this stage = the exclusive scan's total over the scan's range
    G(j.counters)[2 + STAGE] = pos[j.off + j.cap - 1] + flag[j.off + j.cap - 1] - pos[j.off];
  if (i >= j.n || !flag[j.off + i]) return;
  const uint32_t o = pos[j.off + i] - pos[j.off];
  if (STAGE == 1) {
    G(j.mx)[o] = G(j.x)[i];
    G(j.my)[o] = G(j.y)[i];
    G(j.mz)[o] = G(j.z)[i];
    if (j.want_t) {
      float tv = G(j.t)[i];
      if (j.s1.ts_method != MH_TS_NONE) {  // FilterAdjustTimestamps over ALL raw points (it runs before the decimation)
        const float tmin = ord2f(G(j.counters)[0]), tmax = ord2f(G(j.counters)[1]);
        const float dt = j.s1.ts_method == MH_TS_MIDDLE_IS_ZERO ? 0.5f * (tmin + tmax) : tmin;
        tv = (tv - dt) + j.s1.ts_offset;
      }
      G(j.mt)[o] = tv;
    }
    G(j.msrc)[o] = j.src ? G(j.src)[i] : i;
  } else {
    G(j.ix)[o] = G(j.mx)[i];
    G(j.iy)[o] = G(j.my)[i];
    G(j.iz)[o] = G(j.mz)[i];
    if (j.want_t) G(j.it)[o] = G(j.mt)[i];  // out_map's stamps are adjusted already
    G(j.isrc)[o] = G(j.msrc)[i];
  }
}

StageParams stage1_of(const mh_preprocess_params* p, size_t n, bool has_t) {
  StageParams s1{};
  s1.inv_res = p->decim_map_resolution > 0.f ? 1.0f / p->decim_map_resolution : 0.f;
  s1.trunc = p->index_mode == MH_INDEX_TRUNC;
  s1.decimate = (p->decim_map_resolution > 0.f && n >= p->min_points_to_filter) ? 1u : 0u;
  s1.range_on = p->range_max > 0.f ? 1u : 0u;
  s1.sq_min = p->range_min * p->range_min;
  s1.sq_max = p->range_max * p->range_max;
  s1.cx = p->range_center[0]; s1.cy = p->range_center[1]; s1.cz = p->range_center[2];
  s1.bbox_mode = p->bbox_mode;
  for (int a = 0; a < 3; a++) { s1.bmin[a] = p->bbox_min[a]; s1.bmax[a] = p->bbox_max[a]; }
  s1.ts_method = has_t ? p->timestamp_method : MH_TS_NONE;
  s1.ts_offset = p->time_offset;
  s1.method = (uint32_t)p->decim_map_method;
  return s1;
}

mh_status preprocess_batch(size_t n_jobs, const mh_scan* const* raws, const mh_preprocess_params* params, size_t params_stride,
                           mh_scan* const* out_maps, mh_scan* const* out_icps) {
  mh_ctx* lead = raws[0]->ctx;
  MH_TRY(set_device(lead));
  hipStream_t s = lead->stream;
  // pinned staging of the leader: job descriptors up, counters down
  const size_t stage_bytes = n_jobs * sizeof(PpJob) + n_jobs * 8 * sizeof(uint32_t);
  if (lead->h_pp_bytes < stage_bytes) {
    if (lead->h_pp) (void)hipHostFree(lead->h_pp);
    lead->h_pp = nullptr;
    lead->h_pp_bytes = 0;
    MH_HIP(hipHostMalloc((void**)&lead->h_pp, 2 * stage_bytes, hipHostMallocDefault));
    lead->h_pp_bytes = 2 * stage_bytes;
  }
  PpJob* h_jobs = reinterpret_cast<PpJob*>(lead->h_pp);
  uint32_t* h_counts = reinterpret_cast<uint32_t*>(lead->h_pp + n_jobs * sizeof(PpJob));
  size_t total = 0, max_n = 0, max_t = 0, table_entries = 0;
  bool any_icp = false, any_t = false;
  for (size_t k = 0; k < n_jobs; k++) {
    total += (raws[k]->n + 63) / 64 * 64;
    uint64_t tsize = 64;
    while (tsize < 2ull * raws[k]->n) tsize <<= 1;
    table_entries += 2 * tsize;  // (both stages; upper bound)
  }
  MH_REQUIRE(total < 0x7FFFFFF0ull, "scans too large for one batch");
  // decimation tables of all scans: keys | first index, in the leader's scratch
  MH_TRY(lead->build_a.reserve(table_entries * sizeof(unsigned long long)));
  MH_TRY(lead->build_b.reserve(table_entries * sizeof(uint32_t)));
  size_t table_off = 0;
  MH_TRY(lead->build_c.reserve(2 * (total ? total : 1) * sizeof(uint32_t)));              // flag | pos
  MH_TRY(lead->build_e.reserve(n_jobs * (sizeof(PpJob) + 8 * sizeof(uint32_t)) + 256));   // descriptors | counters
  PpJob* d_jobs = lead->build_e.as<PpJob>();
  uint32_t* d_counts = reinterpret_cast<uint32_t*>(lead->build_e.as<char>() + (n_jobs * sizeof(PpJob) + 63) / 64 * 64);
  uint32_t* flag = lead->build_c.as<uint32_t>();
  uint32_t* pos = flag + total;
  size_t off = 0;
  for (size_t k = 0; k < n_jobs; k++) {
    const mh_scan* raw = raws[k];
    const mh_preprocess_params* p = reinterpret_cast<const mh_preprocess_params*>(reinterpret_cast<const char*>(params) + k * params_stride);
    mh_scan* om = out_maps[k];
    mh_scan* oi = out_icps ? out_icps[k] : nullptr;
    mh_ctx* ctx = raw->ctx;
    const size_t n = raw->n;
    const bool has_t = raw->t != nullptr;
    PpJob& j = h_jobs[k];
    memset(&j, 0, sizeof(j));
    j.x = raw->x; j.y = raw->y; j.z = raw->z; j.t = raw->t; j.src = raw->src;
    j.n = (uint32_t)n;
    j.off = (uint32_t)off;
    j.cap = (uint32_t)((n + 63) / 64 * 64);
    off += j.cap;
    j.s1 = stage1_of(p, n, has_t);
    j.s2.inv_res = p->decim_icp_resolution > 0.f ? 1.0f / p->decim_icp_resolution : 0.f;
    j.s2.trunc = j.s1.trunc;
    j.s2.decimate = p->decim_icp_resolution > 0.f ? 1u : 0u;  // && stage 1 left >= min_points: decided on the device
    j.s2.ts_method = MH_TS_NONE;
    j.s2.method = (uint32_t)p->decim_icp_method;
    j.min_points = p->min_points_to_filter;
    j.want_icp = oi ? 1u : 0u;
    j.want_t = has_t ? 1u : 0u;
    uint64_t tsize = 64;
    while (tsize < 2ull * n) tsize <<= 1;
    j.tsize1 = j.s1.decimate ? (uint32_t)tsize : 0u;
    j.tsize2 = (oi && j.s2.decimate && n >= p->min_points_to_filter) ? (uint32_t)tsize : 0u;
    if (!j.tsize2) j.s2.decimate = 0;
    j.keys1 = lead->build_a.as<unsigned long long>() + table_off;
    j.keys2 = j.keys1 + j.tsize1;
    j.first1 = lead->build_b.as<uint32_t>() + table_off;
    j.first2 = j.first1 + j.tsize1;
    table_off += (size_t)j.tsize1 + j.tsize2;
    j.counters = d_counts + 8 * k;
    MH_TRY(scan_alloc(om, n, has_t, true));  // capacity: the raw size; the real counts arrive with the read-back below
    j.mx = (float*)om->x; j.my = (float*)om->y; j.mz = (float*)om->z; j.mt = (float*)om->t; j.msrc = (uint32_t*)om->src;
    if (oi) {
      MH_TRY(scan_alloc(oi, n, has_t, true));
      j.ix = (float*)oi->x; j.iy = (float*)oi->y; j.iz = (float*)oi->z; j.it = (float*)oi->t; j.isrc = (uint32_t*)oi->src;
      any_icp = true;
    }
    any_t = any_t || (has_t && p->timestamp_method != MH_TS_NONE);
    max_n = n > max_n ? n : max_n;
    const size_t ts = j.tsize1 > j.tsize2 ? j.tsize1 : j.tsize2;
    max_t = ts > max_t ? ts : max_t;
    // uploads / earlier work still queued on this scan's own stream come first
    if (ctx != lead && ctx->stream != s && hipStreamQuery(ctx->stream) != hipSuccess) {
      MH_HIP(hipEventRecord(ctx->ev_ready, ctx->stream));
      MH_HIP(hipStreamWaitEvent(s, ctx->ev_ready, 0));
    }
  }
  (void)hipGetLastError();  // hipStreamQuery's hipErrorNotReady is not an error
  if (max_n == 0) {
    for (size_t k = 0; k < n_jobs; k++) {
      out_maps[k]->n = 0;
      if (out_icps && out_icps[k]) out_icps[k]->n = 0;
    }
    return MH_OK;
  }
  size_t tmp = 0;
  MH_HIP(rocprim::exclusive_scan(nullptr, tmp, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, total, rocprim::plus<uint32_t>(), s));
  // ClosestToAverage stages: one scan at a time through a stable sort by voxel key (scratch: keys | sorted keys | indices | order)
  bool any_cta = false;
  for (size_t k = 0; k < n_jobs; k++) any_cta = any_cta || h_jobs[k].s1.method == MH_DECIMATE_CLOSEST_TO_AVERAGE || h_jobs[k].s2.method == MH_DECIMATE_CLOSEST_TO_AVERAGE;
  unsigned long long *cta_keys = nullptr, *cta_keys_s = nullptr;
  uint32_t *cta_idx = nullptr, *cta_perm = nullptr;
  if (any_cta) {
    MH_TRY(lead->build_d.reserve(2 * max_n * sizeof(unsigned long long) + 2 * max_n * sizeof(uint32_t) + 256));
    cta_keys = lead->build_d.as<unsigned long long>();
    cta_keys_s = cta_keys + max_n;
    cta_idx = reinterpret_cast<uint32_t*>(cta_keys_s + max_n);
    cta_perm = cta_idx + max_n;
    size_t t2 = 0;
    MH_HIP(rocprim::radix_sort_pairs(nullptr, t2, cta_keys, cta_keys_s, cta_idx, cta_perm, (uint32_t)max_n, 0, 64, s));
    if (t2 > tmp) tmp = t2;
  }
  MH_TRY(lead->sort_tmp.reserve(tmp));
  auto closest_to_average = [&](int stage) -> mh_status {
    if (!any_cta) return MH_OK;
    for (size_t k = 0; k < n_jobs; k++) {
      const PpJob& j = h_jobs[k];
      const StageParams& sp = stage == 1 ? j.s1 : j.s2;
      if (sp.method != MH_DECIMATE_CLOSEST_TO_AVERAGE || !sp.decimate || !j.n || (stage == 2 && !j.want_icp)) continue;
      const uint32_t N = j.n;
      const dim3 g1(nblk(N, 256));
      if (stage == 1) hipLaunchKernelGGL(k_cta_keys<1>, g1, dim3(256), 0, s, d_jobs, (uint32_t)k, cta_keys, cta_idx);
      else hipLaunchKernelGGL(k_cta_keys<2>, g1, dim3(256), 0, s, d_jobs, (uint32_t)k, cta_keys, cta_idx);
      size_t tb2 = lead->sort_tmp.bytes;
      MH_HIP(rocprim::radix_sort_pairs(lead->sort_tmp.p, tb2, cta_keys, cta_keys_s, cta_idx, cta_perm, N, 0, 64, s));
      if (stage == 1) hipLaunchKernelGGL(k_cta_choose<1>, g1, dim3(256), 0, s, d_jobs, (uint32_t)k, cta_keys_s, cta_perm);
      else hipLaunchKernelGGL(k_cta_choose<2>, g1, dim3(256), 0, s, d_jobs, (uint32_t)k, cta_keys_s, cta_perm);
    }
    return MH_OK;
  };
  MH_HIP(hipMemcpyAsync(d_jobs, h_jobs, n_jobs * sizeof(PpJob), hipMemcpyHostToDevice, s));
  const uint32_t B = 256;
  const dim3 grid(nblk(max_n, B), (uint32_t)n_jobs);
  const uint32_t init_blocks = nblk(max_t ? max_t : 8, B);
  hipLaunchKernelGGL(k_pp_init_b, dim3(init_blocks < 512u ? init_blocks : 512u, (uint32_t)n_jobs), dim3(B), 0, s, d_jobs);
  if (any_t) hipLaunchKernelGGL(k_pp_tminmax_b, dim3(grid.x < 128u ? grid.x : 128u, (uint32_t)n_jobs), dim3(B), 0, s, d_jobs);
  hipLaunchKernelGGL(k_pp_insert_b<1>, grid, dim3(B), 0, s, d_jobs);
  MH_TRY(closest_to_average(1));
  hipLaunchKernelGGL(k_pp_flag_b<1>, grid, dim3(B), 0, s, d_jobs, flag);
  size_t tb = lead->sort_tmp.bytes;
  MH_HIP(rocprim::exclusive_scan(lead->sort_tmp.p, tb, flag, pos, 0u, total, rocprim::plus<uint32_t>(), s));
  hipLaunchKernelGGL(k_pp_compact_b<1>, grid, dim3(B), 0, s, d_jobs, flag, pos);
  if (any_icp) {
    hipLaunchKernelGGL(k_pp_insert_b<2>, grid, dim3(B), 0, s, d_jobs);
    MH_TRY(closest_to_average(2));
    hipLaunchKernelGGL(k_pp_flag_b<2>, grid, dim3(B), 0, s, d_jobs, flag);
    tb = lead->sort_tmp.bytes;
    MH_HIP(rocprim::exclusive_scan(lead->sort_tmp.p, tb, flag, pos, 0u, total, rocprim::plus<uint32_t>(), s));
    hipLaunchKernelGGL(k_pp_compact_b<2>, grid, dim3(B), 0, s, d_jobs, flag, pos);
  }
  MH_HIP(hipGetLastError());
  MH_HIP(hipMemcpyAsync(h_counts, d_counts, n_jobs * 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  MH_HIP(mh::wait_stream(s));
  bool out_of_range = false;
  for (size_t k = 0; k < n_jobs; k++) {
    const uint32_t* c = h_counts + 8 * k;
    out_of_range = out_of_range || (c[2] & 1u);
    out_maps[k]->n = c[3];
    if (out_icps && out_icps[k]) out_icps[k]->n = c[4];
  }
  if (out_of_range)
    return fail(MH_ERR_OUT_OF_RANGE, "a point's decimation voxel index exceeds the +-2^20 range of the packed key");
  return MH_OK;
}

}  // namespace

extern "C" {

mh_status mh_scan_set_timestamps(mh_scan* scan, const float* t, size_t n, int32_t mem) {
  MH_REQUIRE(scan, "null scan");
  MH_REQUIRE(mem == MH_MEM_HOST || mem == MH_MEM_DEVICE, "bad mem space");
  MH_REQUIRE(n == scan->n, "time stamp count differs from the scan size");
  MH_REQUIRE(n == 0 || t, "null time stamps");
  mh_ctx* ctx = scan->ctx;
  MH_TRY(set_device(ctx));
  const size_t stride = ((n * sizeof(float) + 255) / 256) * 256;
  if (scan->aux.bytes < 2 * stride) {
    MH_HIP(mh::wait_stream(ctx->stream));
    MH_TRY(scan->aux.reserve(2 * stride ? 2 * stride : 256));
  }
  if (n) {
    MH_HIP(hipMemcpyAsync(scan->aux.p, t, n * sizeof(float),
                          mem == MH_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, ctx->stream));
    if (mem == MH_MEM_HOST) MH_HIP(mh::wait_stream(ctx->stream));  // host array is borrowed for the call only
  }
  scan->t = (const float*)scan->aux.p;
  return MH_OK;
}

mh_status mh_scan_update_aos(mh_scan* scan, const void* data, size_t n, size_t point_step, size_t off_x, size_t off_y,
                             size_t off_z, int64_t off_t, int32_t mem) {
  MH_REQUIRE(scan, "null scan");
  MH_REQUIRE(mem == MH_MEM_HOST || mem == MH_MEM_DEVICE || mem == MH_MEM_HOST_PINNED, "bad mem space");
  MH_REQUIRE(n == 0 || data, "null data");
  MH_REQUIRE(n < 0x7FFFFFFFull, "scan too large");
  MH_REQUIRE(point_step >= 12 && point_step % 4 == 0 && point_step <= (1u << 16), "point_step must be a multiple of 4 in [12, 65536]");
  MH_REQUIRE(off_x % 4 == 0 && off_y % 4 == 0 && off_z % 4 == 0 && off_x + 4 <= point_step && off_y + 4 <= point_step &&
                 off_z + 4 <= point_step, "coordinate offsets must be multiples of 4 inside the record");
  MH_REQUIRE(off_t < 0 || (off_t % 4 == 0 && (size_t)off_t + 4 <= point_step), "time stamp offset must be a multiple of 4 inside the record");
  mh_ctx* ctx = scan->ctx;
  MH_TRY(set_device(ctx));
  hipStream_t s = ctx->stream;
  const size_t stride = ((n * sizeof(float) + 255) / 256) * 256;
  const size_t raw_bytes = n * point_step;
  if (scan->xyz.bytes < 3 * stride || (off_t >= 0 && scan->aux.bytes < 2 * stride) || ctx->build_a.bytes < raw_bytes) {
    MH_HIP(mh::wait_stream(s));  // nobody may still read the old buffers
    MH_TRY(scan->xyz.reserve(3 * stride ? 3 * stride : 256));
    if (off_t >= 0) MH_TRY(scan->aux.reserve(2 * stride ? 2 * stride : 256));
    MH_TRY(ctx->build_a.reserve(raw_bytes ? raw_bytes : 256));
  }
  char* base = scan->xyz.as<char>();
  scan->x = (const float*)base;
  scan->y = (const float*)(base + stride);
  scan->z = (const float*)(base + 2 * stride);
  scan->t = off_t >= 0 ? (const float*)scan->aux.p : nullptr;
  scan->src = nullptr;
  scan->n = n;
  scan_drop_tiles(scan);
  if (!n) return MH_OK;
  const uint32_t* recs = (const uint32_t*)data;
  if (mem != MH_MEM_DEVICE) {  // the only copy of the call: the raw bytes as they are
    MH_HIP(hipMemcpyAsync(ctx->build_a.p, data, raw_bytes, hipMemcpyHostToDevice, s));
    recs = ctx->build_a.as<uint32_t>();
  }
  hipLaunchKernelGGL(k_pp_deinterleave, dim3(nblk(n, 256)), dim3(256), 0, s, recs, (uint32_t)n, (uint32_t)(point_step / 4),
                     (uint32_t)(off_x / 4), (uint32_t)(off_y / 4), (uint32_t)(off_z / 4), off_t >= 0 ? (int32_t)(off_t / 4) : -1,
                     (float*)scan->x, (float*)scan->y, (float*)scan->z, (float*)scan->t);
  MH_HIP(hipGetLastError());
  if (mem != MH_MEM_HOST_PINNED) MH_HIP(mh::wait_stream(s));  // `data` is borrowed for the call only
  return MH_OK;
}

static mh_status check_preprocess_args(const mh_scan* raw, const mh_preprocess_params* p, const mh_scan* out_map, const mh_scan* out_icp) {
  MH_REQUIRE(raw && p && out_map, "null argument");
  MH_REQUIRE(out_map != raw && out_icp != raw && out_map != out_icp, "outputs must be distinct scans");
  MH_REQUIRE(out_map->ctx == raw->ctx && (!out_icp || out_icp->ctx == raw->ctx), "scans belong to different contexts");
  MH_REQUIRE(p->decim_map_resolution >= 0.f && p->decim_icp_resolution >= 0.f, "negative decimation resolution");
  MH_REQUIRE(p->index_mode == MH_INDEX_FLOOR || p->index_mode == MH_INDEX_TRUNC, "bad index_mode");
  MH_REQUIRE(p->bbox_mode >= MH_BBOX_OFF && p->bbox_mode <= MH_BBOX_KEEP_INSIDE, "bad bbox_mode");
  MH_REQUIRE(p->timestamp_method >= MH_TS_NONE && p->timestamp_method <= MH_TS_EARLIEST_IS_ZERO, "bad timestamp_method");
  MH_REQUIRE((p->decim_map_method == MH_DECIMATE_FIRST_POINT || p->decim_map_method == MH_DECIMATE_CLOSEST_TO_AVERAGE) &&
                 (p->decim_icp_method == MH_DECIMATE_FIRST_POINT || p->decim_icp_method == MH_DECIMATE_CLOSEST_TO_AVERAGE),
             "bad decimate method");
  MH_REQUIRE(raw->n < 0x7FFFFFF0ull, "scan too large");
  return MH_OK;
}

mh_status mh_scan_preprocess(const mh_scan* raw, const mh_preprocess_params* p, mh_scan* out_map, mh_scan* out_icp) {
  MH_TRY(check_preprocess_args(raw, p, out_map, out_icp));
  return preprocess_batch(1, &raw, p, sizeof(mh_preprocess_params), &out_map, out_icp ? &out_icp : nullptr);
}

mh_status mh_scan_preprocess_batch(size_t n_jobs, const mh_scan* const* raws, const mh_preprocess_params* params,
                                   size_t params_stride, mh_scan* const* out_maps, mh_scan* const* out_icps) {
  MH_REQUIRE(n_jobs == 0 || (raws && params && out_maps), "null argument");
  MH_REQUIRE(params_stride == 0 || params_stride >= sizeof(mh_preprocess_params), "params_stride smaller than the structure");
  if (!n_jobs) return MH_OK;
  for (size_t k = 0; k < n_jobs; k++) {
    const mh_preprocess_params* p = reinterpret_cast<const mh_preprocess_params*>(reinterpret_cast<const char*>(params) + k * params_stride);
    MH_TRY(check_preprocess_args(raws[k], p, out_maps[k], out_icps ? out_icps[k] : nullptr));
    MH_REQUIRE(raws[k]->ctx->device == raws[0]->ctx->device, "scans live on different devices");
    for (size_t q = 0; q < k; q++) {
      const mh_scan* others[3] = {raws[q], out_maps[q], out_icps ? out_icps[q] : nullptr};
      for (const mh_scan* o : others)
        MH_REQUIRE(!o || (o != out_maps[k] && (!out_icps || o != out_icps[k])), "an output scan appears twice in the batch");
    }
  }
  return preprocess_batch(n_jobs, raws, params, params_stride, out_maps, out_icps);
}

mh_status mh_scan_deskew(const mh_scan* in, const double twist[6], mh_scan* out) {
  MH_REQUIRE(in && out, "null argument");
  MH_REQUIRE(in != out, "`out` must differ from `in`");
  MH_REQUIRE(in->ctx->device == out->ctx->device, "scans live on different devices");
  // `in` may belong to another context of the same device (a layer prepared on a second stream, complete by now):
  // the work is ordered on `out`'s stream
  mh_ctx* ctx = out->ctx;
  MH_TRY(set_device(ctx));
  hipStream_t s = ctx->stream;
  const size_t n = in->n;
  MH_TRY(scan_alloc(out, n, in->t != nullptr, in->src != nullptr));
  if (!n) return MH_OK;
  if (twist && in->t) {
    Twist tw;
    for (int i = 0; i < 6; i++) {
      MH_REQUIRE(isfinite(twist[i]), "non-finite twist");
      tw.v[i] = twist[i];
    }
    hipLaunchKernelGGL(k_pp_deskew, dim3(nblk(n, 256)), dim3(256), 0, s, in->x, in->y, in->z, in->t, in->src,
                       (uint32_t)n, tw, (float*)out->x, (float*)out->y, (float*)out->z, (float*)out->t, (uint32_t*)out->src);
    MH_HIP(hipGetLastError());
  } else {  // skip_deskew / silently_ignore_no_timestamps
    if (in->t) MH_HIP(hipMemcpyAsync((void*)out->t, in->t, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (in->src) MH_HIP(hipMemcpyAsync((void*)out->src, in->src, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    MH_HIP(hipMemcpyAsync((void*)out->x, in->x, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    MH_HIP(hipMemcpyAsync((void*)out->y, in->y, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    MH_HIP(hipMemcpyAsync((void*)out->z, in->z, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  return MH_OK;
}

static void bbox_from_words(const uint32_t h[8], float bb_min[3], float bb_max[3], uint64_t* n_finite) {
  auto ord = [](uint32_t u) { u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u; float f; memcpy(&f, &u, 4); return f; };
  for (int a = 0; a < 3; a++) {
    bb_min[a] = h[6] ? ord(h[a]) : 0.f;
    bb_max[a] = h[6] ? ord(h[3 + a]) : 0.f;
  }
  if (n_finite) *n_finite = h[6];
}

mh_status mh_scan_deskew_pair(const mh_scan* in_a, const mh_scan* in_b, const double twist[6], mh_scan* out_a, mh_scan* out_b,
                              float bb_min[3], float bb_max[3], uint64_t* n_finite) {
  MH_REQUIRE(in_a && in_b && out_a && out_b && bb_min && bb_max, "null argument");
  MH_REQUIRE(in_a != out_a && in_b != out_b && out_a != out_b && in_a != out_b && in_b != out_a, "the four scans must be distinct");
  MH_REQUIRE(out_a->ctx == out_b->ctx, "the outputs belong to different contexts");
  mh_ctx* ctx = out_a->ctx;
  const bool fused = twist && in_a->t && in_b->t && in_a->n && in_b->n && in_b->n <= 65536 && in_a->n < 0x7FFFFF00ull &&
                     in_a->ctx->device == ctx->device && in_b->ctx->device == ctx->device;
  if (!fused) {  // copies (skip_deskew / no time stamps), empty or huge layers: the separate calls
    MH_TRY(mh_scan_deskew(in_a, twist, out_a));
    MH_TRY(mh_scan_deskew(in_b, twist, out_b));
    return mh_scan_bbox(out_b, bb_min, bb_max, n_finite);
  }
  MH_TRY(set_device(ctx));
  hipStream_t s = ctx->stream;
  Twist tw;
  for (int i = 0; i < 6; i++) {
    MH_REQUIRE(isfinite(twist[i]), "non-finite twist");
    tw.v[i] = twist[i];
  }
  MH_TRY(scan_alloc(out_a, in_a->n, true, in_a->src != nullptr));
  MH_TRY(scan_alloc(out_b, in_b->n, true, in_b->src != nullptr));
  if (!ctx->h_small) MH_HIP(hipHostMalloc((void**)&ctx->h_small, 64 * sizeof(uint32_t), hipHostMallocDefault));
  auto layer = [](const mh_scan* in, mh_scan* out) {
    DeskewLayer l;
    l.x = in->x; l.y = in->y; l.z = in->z; l.t = in->t; l.src = in->src;
    l.ox = (float*)out->x; l.oy = (float*)out->y; l.oz = (float*)out->z; l.ot = (float*)out->t; l.osrc = (uint32_t*)out->src;
    l.n = (uint32_t)in->n;
    return l;
  };
  hipLaunchKernelGGL(k_pp_deskew_pair, dim3(1 + nblk(in_a->n, 1024)), dim3(1024), 0, s, layer(in_a, out_a), layer(in_b, out_b), tw,
                     ctx->h_small);
  MH_HIP(hipGetLastError());
  MH_HIP(mh::wait_stream(s));
  uint32_t h[8];
  for (int a = 0; a < 7; a++) h[a] = ctx->h_small[a];
  bbox_from_words(h, bb_min, bb_max, n_finite);
  return MH_OK;
}

mh_status mh_scan_bbox(const mh_scan* scan, float bb_min[3], float bb_max[3], uint64_t* n_finite) {
  MH_REQUIRE(scan && bb_min && bb_max, "null argument");
  mh_ctx* ctx = scan->ctx;
  MH_TRY(set_device(ctx));
  hipStream_t s = ctx->stream;
  uint32_t h[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
  if (scan->n && scan->n <= 65536) {
    if (!ctx->h_small) MH_HIP(hipHostMalloc((void**)&ctx->h_small, 64 * sizeof(uint32_t), hipHostMallocDefault));
    hipLaunchKernelGGL(k_pp_bbox_one, dim3(1), dim3(1024), 0, s, scan->x, scan->y, scan->z, (uint32_t)scan->n, ctx->h_small);
    MH_HIP(hipGetLastError());
    MH_HIP(mh::wait_stream(s));
    for (int a = 0; a < 7; a++) h[a] = ctx->h_small[a];
  } else if (scan->n) {
    MH_TRY(ctx->build_e.reserve(64));
    uint32_t* counters = ctx->build_e.as<uint32_t>();
    MH_HIP(hipMemcpyAsync(counters, h, sizeof(h), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_pp_bbox, dim3(nblk(scan->n, 256) < 128u ? nblk(scan->n, 256) : 128u), dim3(256), 0, s, scan->x,
                       scan->y, scan->z, (uint32_t)scan->n, counters);
    MH_HIP(hipGetLastError());
    MH_HIP(hipMemcpyAsync(h, counters, sizeof(h), hipMemcpyDeviceToHost, s));
    MH_HIP(mh::wait_stream(s));
  }
  for (int a = 0; a < 3; a++) {
    bb_min[a] = h[6] ? [](uint32_t u) { u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u; float f; memcpy(&f, &u, 4); return f; }(h[a]) : 0.f;
    bb_max[a] = h[6] ? [](uint32_t u) { u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u; float f; memcpy(&f, &u, 4); return f; }(h[3 + a]) : 0.f;
  }
  if (n_finite) *n_finite = h[6];
  return MH_OK;
}

mh_status mh_scan_download(const mh_scan* scan, float* x, float* y, float* z, float* t, uint32_t* src_idx) {
  MH_REQUIRE(scan, "null scan");
  mh_ctx* ctx = scan->ctx;
  MH_TRY(set_device(ctx));
  hipStream_t s = ctx->stream;
  const size_t n = scan->n;
  if (n) {
    if (x) MH_HIP(hipMemcpyAsync(x, scan->x, n * sizeof(float), hipMemcpyDeviceToHost, s));
    if (y) MH_HIP(hipMemcpyAsync(y, scan->y, n * sizeof(float), hipMemcpyDeviceToHost, s));
    if (z) MH_HIP(hipMemcpyAsync(z, scan->z, n * sizeof(float), hipMemcpyDeviceToHost, s));
    if (t) {
      if (scan->t) MH_HIP(hipMemcpyAsync(t, scan->t, n * sizeof(float), hipMemcpyDeviceToHost, s));
      else memset(t, 0, n * sizeof(float));
    }
    if (src_idx) {
      if (scan->src) MH_HIP(hipMemcpyAsync(src_idx, scan->src, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      else memset(src_idx, 0, n * sizeof(uint32_t));
    }
  }
  MH_HIP(mh::wait_stream(s));
  return MH_OK;
}

}  // extern "C"
