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

__global__ void k_pp_deskew(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                            const float* __restrict__ t, const uint32_t* __restrict__ src, uint32_t n, Twist tw,
                            float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz,
                            float* __restrict__ ot, uint32_t* __restrict__ osrc) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float tf = t[i];
  ot[i] = tf;
  if (src) osrc[i] = src[i];
  const double dt = (double)tf;
  const double xi[6] = {0.0, 0.0, 0.0, tw.v[3] * dt, tw.v[4] * dt, tw.v[5] * dt};
  Pose p = se3_exp(xi);  // zero translation part: pure Exp_SO3(w dt)
  p.t(0) = tw.v[0] * dt;
  p.t(1) = tw.v[1] * dt;
  p.t(2) = tw.v[2] * dt;
  float gx, gy, gz;
  transform_point(p.m, x[i], y[i], z[i], gx, gy, gz);
  ox[i] = gx;
  oy[i] = gy;
  oz[i] = gz;
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

// One stage: [decimate] + predicates over `in` -> `out` (compacted, order preserving).  Returns the survivor count.
mh_status run_stage(mh_ctx* ctx, const mh_scan* in, const StageParams& sp, uint32_t* counters, uint32_t count_slot,
                    mh_scan* out, bool want_t) {
  hipStream_t s = ctx->stream;
  const uint32_t N = (uint32_t)in->n;
  const uint32_t B = 256;
  if (N == 0) return scan_alloc(out, 0, want_t, true);
  uint32_t* flag = ctx->build_c.as<uint32_t>();
  uint32_t* pos = flag + N;
  unsigned long long* keys = nullptr;
  uint32_t* first_idx = nullptr;
  uint32_t mask = 0;
  if (sp.decimate) {
    uint64_t tsize = 64;
    while (tsize < 2ull * N) tsize <<= 1;
    MH_TRY(ctx->build_a.reserve(tsize * sizeof(unsigned long long)));
    MH_TRY(ctx->build_b.reserve(tsize * sizeof(uint32_t)));
    keys = ctx->build_a.as<unsigned long long>();
    first_idx = ctx->build_b.as<uint32_t>();
    mask = (uint32_t)(tsize - 1);
    MH_HIP(hipMemsetAsync(keys, 0xFF, tsize * sizeof(unsigned long long), s));
    MH_HIP(hipMemsetAsync(first_idx, 0xFF, tsize * sizeof(uint32_t), s));
    hipLaunchKernelGGL(k_pp_insert, dim3(nblk(N, B)), dim3(B), 0, s, in->x, in->y, in->z, N, sp.inv_res, sp.trunc, keys,
                       first_idx, mask, counters);
  }
  hipLaunchKernelGGL(k_pp_flag, dim3(nblk(N, B)), dim3(B), 0, s, in->x, in->y, in->z, N, sp, keys, first_idx, mask,
                     counters, flag, count_slot);
  size_t tb = ctx->sort_tmp.bytes;
  MH_HIP(rocprim::exclusive_scan(ctx->sort_tmp.p, tb, flag, pos, 0u, N, rocprim::plus<uint32_t>(), s));
  uint32_t h[3] = {0, 0, 0};  // range flag, stage-1 count, stage-2 count: one read-back per stage
  MH_HIP(hipMemcpyAsync(h, counters + 2, sizeof(h), hipMemcpyDeviceToHost, s));
  MH_HIP(mh::wait_stream(s));
  if (h[0] & 1u)
    return fail(MH_ERR_OUT_OF_RANGE, "a point's decimation voxel index exceeds the +-2^20 range of the packed key");
  const uint32_t M = h[count_slot - 2];
  MH_TRY(scan_alloc(out, M, want_t, true));
  if (M)
    hipLaunchKernelGGL(k_pp_compact, dim3(nblk(N, B)), dim3(B), 0, s, in->x, in->y, in->z, want_t ? in->t : nullptr,
                       in->src, N, flag, pos, sp.ts_method, sp.ts_offset, counters, (float*)out->x, (float*)out->y,
                       (float*)out->z, want_t ? (float*)out->t : nullptr, (uint32_t*)out->src);
  MH_HIP(hipGetLastError());
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

mh_status mh_scan_preprocess(const mh_scan* raw, const mh_preprocess_params* p, mh_scan* out_map, mh_scan* out_icp) {
  MH_REQUIRE(raw && p && out_map, "null argument");
  MH_REQUIRE(out_map != raw && out_icp != raw && out_map != out_icp, "outputs must be distinct scans");
  MH_REQUIRE(out_map->ctx == raw->ctx && (!out_icp || out_icp->ctx == raw->ctx), "scans belong to different contexts");
  MH_REQUIRE(p->decim_map_resolution >= 0.f && p->decim_icp_resolution >= 0.f, "negative decimation resolution");
  MH_REQUIRE(p->index_mode == MH_INDEX_FLOOR || p->index_mode == MH_INDEX_TRUNC, "bad index_mode");
  MH_REQUIRE(p->bbox_mode >= MH_BBOX_OFF && p->bbox_mode <= MH_BBOX_KEEP_INSIDE, "bad bbox_mode");
  MH_REQUIRE(p->timestamp_method >= MH_TS_NONE && p->timestamp_method <= MH_TS_EARLIEST_IS_ZERO, "bad timestamp_method");
  MH_REQUIRE(raw->n < 0x7FFFFFF0ull, "scan too large");
  mh_ctx* ctx = raw->ctx;
  MH_TRY(set_device(ctx));
  hipStream_t s = ctx->stream;
  const size_t n = raw->n;
  const bool has_t = raw->t != nullptr;

  MH_TRY(ctx->build_c.reserve(2 * (n ? n : 1) * sizeof(uint32_t)));  // flag | pos
  MH_TRY(ctx->build_e.reserve(64));  // counters: tmin, tmax, range flag, survivors of stage 1, of stage 2
  uint32_t* counters = ctx->build_e.as<uint32_t>();
  const uint32_t init[5] = {0xFFFFFFFFu, 0u, 0u, 0u, 0u};
  MH_HIP(hipMemcpyAsync(counters, init, sizeof(init), hipMemcpyHostToDevice, s));
  if (n) {
    size_t tmp = 0;
    MH_HIP(rocprim::exclusive_scan(nullptr, tmp, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, n, rocprim::plus<uint32_t>(), s));
    MH_TRY(ctx->sort_tmp.reserve(tmp));
  }
  if (has_t && n && p->timestamp_method != MH_TS_NONE)
    hipLaunchKernelGGL(k_pp_tminmax, dim3(nblk(n, 256) < 128u ? nblk(n, 256) : 128u), dim3(256), 0, s, raw->t, (uint32_t)n, counters);

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
  MH_TRY(run_stage(ctx, raw, s1, counters, 3, out_map, has_t));

  if (out_icp) {
    StageParams s2{};
    s2.inv_res = p->decim_icp_resolution > 0.f ? 1.0f / p->decim_icp_resolution : 0.f;
    s2.trunc = s1.trunc;
    s2.decimate = (p->decim_icp_resolution > 0.f && out_map->n >= p->min_points_to_filter) ? 1u : 0u;
    s2.ts_method = MH_TS_NONE;  // out_map's stamps are adjusted already
    MH_TRY(run_stage(ctx, out_map, s2, counters, 4, out_icp, has_t));
  }
  return MH_OK;
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
