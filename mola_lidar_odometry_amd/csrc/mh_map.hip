// mh_map.hip -- device-resident local map: the NN-search target that stands in for
// mola::HashedVoxelPointCloud [U] (lidar3d-default.yaml:228-242) in its NearestNeighborsCapable role.
//
// Build = clear() + insertPoint() for every point in order, restated data-parallel:
//   key_i   = packed voxel index of point i                 (coordToGlobalIdx [U], SURVEY App.A)
//   stable radix sort of (key_i, i)                          -> voxel-contiguous, in-voxel insertion order
//   rank_i  = position of i inside its voxel run             -> keep iff rank_i < max_points_per_voxel
//   compact kept points into 16-byte records {x,y,z,src}     (one dwordx4 per candidate in the NN kernel)
//   one hash-table slot {key, first, count} per voxel        (open addressing, load factor <= 0.5)
// The sort/scans use rocPRIM device primitives (build is per key-frame, not per ICP iteration); the
// kernels around them are hand-written.
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <new>
#include <vector>

#include "mh_internal.h"

using namespace mh;

namespace {

__device__ __forceinline__ int voxel_index(float c, float inv_vs, uint32_t trunc) {
  const float s = c * inv_vs;  // fp32 product, like the reference
  return trunc ? (int)s : (int)floorf(s);
}

__global__ void k_keys(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, uint32_t n,
                       float inv_vs, uint32_t trunc, unsigned long long* __restrict__ keys, uint32_t* __restrict__ idx,
                       uint32_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float px = x[i], py = y[i], pz = z[i];
  unsigned long long k = kEmptyKey;
  if (isfinite(px) && isfinite(py) && isfinite(pz)) {
    const float sx = px * inv_vs, sy = py * inv_vs, sz = pz * inv_vs;
    if (fabsf(sx) < 1.0e6f && fabsf(sy) < 1.0e6f && fabsf(sz) < 1.0e6f) {
      k = pack_key(voxel_index(px, inv_vs, trunc), voxel_index(py, inv_vs, trunc), voxel_index(pz, inv_vs, trunc));
    } else {
      atomicOr(&flags[0], 1u);  // voxel index does not fit 21 bits
    }
  }
  keys[i] = k;
  idx[i] = i;
}

// head[i] = 1 where a new voxel run starts; counters[1] = number of valid (finite) points
__global__ void k_heads(const unsigned long long* __restrict__ ks, uint32_t n, uint32_t* __restrict__ head,
                        uint32_t* __restrict__ counters) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = ks[i];
  const bool valid = k != kEmptyKey;
  head[i] = (valid && (i == 0 || ks[i - 1] != k)) ? 1u : 0u;
  if (valid && (i + 1 == n || ks[i + 1] == kEmptyKey)) counters[1] = i + 1;
}

__global__ void k_vstart(const uint32_t* __restrict__ head, const uint32_t* __restrict__ vid1, uint32_t n,
                         uint32_t* __restrict__ vstart) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (head[i]) vstart[vid1[i] - 1] = i;
}

__global__ void k_keep(const unsigned long long* __restrict__ ks, const uint32_t* __restrict__ vid1,
                       const uint32_t* __restrict__ vstart, uint32_t n, uint32_t cap, uint32_t* __restrict__ keep) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t k = 0;
  if (ks[i] != kEmptyKey) {
    const uint32_t rank = i - vstart[vid1[i] - 1];
    k = (cap == 0 || rank < cap) ? 1u : 0u;  // insertPoint drops the point when the voxel is full
  }
  keep[i] = k;
}

__device__ __forceinline__ uint32_t f2ord(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ inline float ord2f(uint32_t u) {
  u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

__global__ void k_scatter(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                          const unsigned long long* __restrict__ ks, const uint32_t* __restrict__ idx_s,
                          const uint32_t* __restrict__ head, const uint32_t* __restrict__ vid1,
                          const uint32_t* __restrict__ vstart, const uint32_t* __restrict__ keep,
                          const uint32_t* __restrict__ outpos, uint32_t n, uint32_t cap,
                          const uint32_t* __restrict__ counters, uint32_t n_vox, float4* __restrict__ pts,
                          unsigned long long* __restrict__ vox_keys, uint32_t* __restrict__ vox_first,
                          uint32_t* __restrict__ vox_count, uint32_t* __restrict__ bbox /*6 ordered uints*/) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  float px = 0, py = 0, pz = 0;
  bool kept = false;
  if (i < n && keep[i]) {
    kept = true;
    const uint32_t src = idx_s[i];
    px = x[src]; py = y[src]; pz = z[src];
    pts[outpos[i]] = make_float4(px, py, pz, __uint_as_float(src));
    if (head[i]) {
      const uint32_t v = vid1[i] - 1;
      const uint32_t n_valid = counters[1];
      const uint32_t end = (v + 1 < n_vox) ? vstart[v + 1] : n_valid;
      const uint32_t cnt = end - i;
      vox_keys[v] = ks[i];
      vox_first[v] = outpos[i];
      vox_count[v] = (cap == 0 || cnt < cap) ? cnt : cap;
    }
  }
  // bounding box of the stored points: wave min/max, then one atomic per wave
  uint32_t mn[3] = {kept ? f2ord(px) : 0xFFFFFFFFu, kept ? f2ord(py) : 0xFFFFFFFFu, kept ? f2ord(pz) : 0xFFFFFFFFu};
  uint32_t mx[3] = {kept ? f2ord(px) : 0u, kept ? f2ord(py) : 0u, kept ? f2ord(pz) : 0u};
  for (int off = 32; off > 0; off >>= 1)
    for (int a = 0; a < 3; a++) {
      mn[a] = min(mn[a], (uint32_t)__shfl_xor((int)mn[a], off));
      mx[a] = max(mx[a], (uint32_t)__shfl_xor((int)mx[a], off));
    }
  if ((threadIdx.x & 63) == 0 && mx[0] != 0u)
    for (int a = 0; a < 3; a++) {
      atomicMin(&bbox[a], mn[a]);
      atomicMax(&bbox[3 + a], mx[a]);
    }
}

__global__ void k_table_insert(const unsigned long long* __restrict__ vox_keys, const uint32_t* __restrict__ vox_first,
                               const uint32_t* __restrict__ vox_count, uint32_t n_vox, MapSlot* __restrict__ slots,
                               uint32_t mask) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n_vox) return;
  const unsigned long long key = vox_keys[v];
  uint32_t h = hash_key(key) & mask;
  for (;;) {
    const unsigned long long old = atomicCAS(&slots[h].key, kEmptyKey, key);
    if (old == kEmptyKey) {  // claimed: keys are unique per voxel, so nobody else writes this payload
      slots[h].first = vox_first[v];
      slots[h].count = vox_count[v];
      return;
    }
    h = (h + 1) & mask;
  }
}

inline uint32_t nblk(size_t n, uint32_t b) { return (uint32_t)((n + b - 1) / b); }

}  // namespace

extern "C" {

mh_status mh_map_create(mh_ctx* ctx, const mh_map_params* params, mh_map** out) {
  MH_REQUIRE(ctx && params && out, "null argument");
  *out = nullptr;
  MH_REQUIRE(params->voxel_size > 0.f && isfinite(params->voxel_size), "voxel_size must be > 0");
  MH_REQUIRE(params->index_mode == MH_INDEX_FLOOR || params->index_mode == MH_INDEX_TRUNC, "bad index_mode");
  mh_map* m = new (std::nothrow) mh_map();
  if (!m) return fail(MH_ERR_OUT_OF_MEMORY, "host allocation failed");
  m->ctx = ctx;
  m->params = *params;
  m->inv_vs = 1.0f / params->voxel_size;
  // an empty map still needs a (tiny) all-empty table so that queries are well defined
  mh_status st = mh_map_build(m, nullptr, nullptr, nullptr, 0, MH_MEM_HOST);
  if (st != MH_OK) {
    delete m;
    return st;
  }
  *out = m;
  return MH_OK;
}

mh_status mh_map_destroy(mh_map* m) {
  if (!m) return MH_OK;
  (void)hipSetDevice(m->ctx->device);
  (void)hipStreamSynchronize(m->ctx->stream);
  m->slots.release();
  m->pts.release();
  m->vox_keys.release();
  m->vox_first.release();
  m->vox_count.release();
  delete m;
  return MH_OK;
}

mh_status mh_map_build(mh_map* m, const float* x, const float* y, const float* z, size_t n, int32_t mem) {
  MH_REQUIRE(m, "null map");
  MH_REQUIRE(mem == MH_MEM_HOST || mem == MH_MEM_DEVICE, "bad mem space");
  MH_REQUIRE(n == 0 || (x && y && z), "null point arrays");
  MH_REQUIRE(n < 0x7FFFFFF0ull, "too many points");
  mh_ctx* ctx = m->ctx;
  MH_TRY(set_device(ctx));
  hipStream_t s = ctx->stream;
  MH_HIP(hipStreamSynchronize(s));  // a rebuild invalidates everything queued against the old content

  uint32_t n_vox = 0, n_pts = 0;
  uint32_t h_counters[12] = {0};
  if (n > 0) {
    const float *dx = x, *dy = y, *dz = z;
    if (mem == MH_MEM_HOST) {
      const size_t stride = ((n * sizeof(float) + 255) / 256) * 256;
      MH_TRY(ctx->staging.reserve(3 * stride));
      MH_TRY(stage_in(ctx, ctx->staging, 0, x, n * sizeof(float), mem));
      MH_TRY(stage_in(ctx, ctx->staging, stride, y, n * sizeof(float), mem));
      MH_TRY(stage_in(ctx, ctx->staging, 2 * stride, z, n * sizeof(float), mem));
      dx = (const float*)ctx->staging.as<char>();
      dy = (const float*)(ctx->staging.as<char>() + stride);
      dz = (const float*)(ctx->staging.as<char>() + 2 * stride);
    }
    const uint32_t N = (uint32_t)n;
    // scratch carve-up
    MH_TRY(ctx->build_a.reserve(2 * n * sizeof(unsigned long long)));  // keys in | keys sorted
    MH_TRY(ctx->build_b.reserve(2 * n * sizeof(uint32_t)));            // idx in | idx sorted
    MH_TRY(ctx->build_c.reserve(2 * n * sizeof(uint32_t)));            // head | vid1
    MH_TRY(ctx->build_d.reserve(2 * n * sizeof(uint32_t)));            // keep | outpos
    MH_TRY(ctx->build_e.reserve(n * sizeof(uint32_t) + 64));           // vstart | counters(12)
    unsigned long long* keys = ctx->build_a.as<unsigned long long>();
    unsigned long long* keys_s = keys + n;
    uint32_t* idx = ctx->build_b.as<uint32_t>();
    uint32_t* idx_s = idx + n;
    uint32_t* head = ctx->build_c.as<uint32_t>();
    uint32_t* vid1 = head + n;
    uint32_t* keep = ctx->build_d.as<uint32_t>();
    uint32_t* outpos = keep + n;
    uint32_t* vstart = ctx->build_e.as<uint32_t>();
    uint32_t* counters = vstart + n;  // [0]=range flag [1]=n_valid [2..7]=bbox ordered

    const uint32_t init_counters[12] = {0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0};
    MH_HIP(hipMemcpyAsync(counters, init_counters, sizeof(init_counters), hipMemcpyHostToDevice, s));
    const uint32_t B = 256;
    hipLaunchKernelGGL(k_keys, dim3(nblk(n, B)), dim3(B), 0, s, dx, dy, dz, N, m->inv_vs,
                       (uint32_t)(m->params.index_mode == MH_INDEX_TRUNC), keys, idx, counters);
    size_t tmp = 0;
    MH_HIP(rocprim::radix_sort_pairs(nullptr, tmp, keys, keys_s, idx, idx_s, N, 0, 64, s));
    {
      size_t t2 = 0;
      MH_HIP(rocprim::inclusive_scan(nullptr, t2, head, vid1, N, rocprim::plus<uint32_t>(), s));
      if (t2 > tmp) tmp = t2;
      MH_HIP(rocprim::exclusive_scan(nullptr, t2, keep, outpos, 0u, N, rocprim::plus<uint32_t>(), s));
      if (t2 > tmp) tmp = t2;
    }
    MH_TRY(ctx->sort_tmp.reserve(tmp));
    size_t tb = ctx->sort_tmp.bytes;
    MH_HIP(rocprim::radix_sort_pairs(ctx->sort_tmp.p, tb, keys, keys_s, idx, idx_s, N, 0, 64, s));
    hipLaunchKernelGGL(k_heads, dim3(nblk(n, B)), dim3(B), 0, s, keys_s, N, head, counters);
    tb = ctx->sort_tmp.bytes;
    MH_HIP(rocprim::inclusive_scan(ctx->sort_tmp.p, tb, head, vid1, N, rocprim::plus<uint32_t>(), s));
    hipLaunchKernelGGL(k_vstart, dim3(nblk(n, B)), dim3(B), 0, s, head, vid1, N, vstart);
    hipLaunchKernelGGL(k_keep, dim3(nblk(n, B)), dim3(B), 0, s, keys_s, vid1, vstart, N, m->params.max_points_per_voxel,
                       keep);
    tb = ctx->sort_tmp.bytes;
    MH_HIP(rocprim::exclusive_scan(ctx->sort_tmp.p, tb, keep, outpos, 0u, N, rocprim::plus<uint32_t>(), s));
    // sizes back to the host
    uint32_t h_last[3] = {0, 0, 0};
    MH_HIP(hipMemcpyAsync(&h_last[0], vid1 + (n - 1), 4, hipMemcpyDeviceToHost, s));
    MH_HIP(hipMemcpyAsync(&h_last[1], outpos + (n - 1), 4, hipMemcpyDeviceToHost, s));
    MH_HIP(hipMemcpyAsync(&h_last[2], keep + (n - 1), 4, hipMemcpyDeviceToHost, s));
    MH_HIP(hipMemcpyAsync(h_counters, counters, 8, hipMemcpyDeviceToHost, s));
    MH_HIP(hipStreamSynchronize(s));
    if (h_counters[0] & 1u)
      return fail(MH_ERR_OUT_OF_RANGE, "a point's voxel index exceeds the +-2^20 range of the packed key "
                                       "(|coord|/voxel_size must be < 1e6)");
    n_vox = h_last[0];
    n_pts = h_last[1] + h_last[2];

    MH_TRY(m->pts.reserve((size_t)(n_pts ? n_pts : 1) * sizeof(float4)));
    MH_TRY(m->vox_keys.reserve((size_t)(n_vox ? n_vox : 1) * sizeof(unsigned long long)));
    MH_TRY(m->vox_first.reserve((size_t)(n_vox ? n_vox : 1) * sizeof(uint32_t)));
    MH_TRY(m->vox_count.reserve((size_t)(n_vox ? n_vox : 1) * sizeof(uint32_t)));
    if (n_pts) {
      hipLaunchKernelGGL(k_scatter, dim3(nblk(n, B)), dim3(B), 0, s, dx, dy, dz, keys_s, idx_s, head, vid1, vstart, keep,
                         outpos, N, m->params.max_points_per_voxel, counters, n_vox, m->pts.as<float4>(),
                         m->vox_keys.as<unsigned long long>(), m->vox_first.as<uint32_t>(), m->vox_count.as<uint32_t>(),
                         counters + 2);
      MH_HIP(hipMemcpyAsync(h_counters, counters, sizeof(h_counters), hipMemcpyDeviceToHost, s));
    }
  }
  // hash table: power of two, load factor <= 0.5
  uint64_t tsize = 64;
  while (tsize < 2ull * n_vox) tsize <<= 1;
  MH_TRY(m->slots.reserve(tsize * sizeof(MapSlot)));
  MH_HIP(hipMemsetAsync(m->slots.p, 0xFF, tsize * sizeof(MapSlot), s));
  if (n_vox)
    hipLaunchKernelGGL(k_table_insert, dim3(nblk(n_vox, 256)), dim3(256), 0, s, m->vox_keys.as<unsigned long long>(),
                       m->vox_first.as<uint32_t>(), m->vox_count.as<uint32_t>(), n_vox, m->slots.as<MapSlot>(),
                       (uint32_t)(tsize - 1));
  MH_HIP(hipGetLastError());
  MH_HIP(hipStreamSynchronize(s));
  m->n_points = n_pts;
  m->n_voxels = n_vox;
  m->n_offered = n;
  m->table_size = tsize;
  for (int a = 0; a < 3; a++) {
    m->bbox_min[a] = n_pts ? ord2f(h_counters[2 + a]) : 0.f;
    m->bbox_max[a] = n_pts ? ord2f(h_counters[5 + a]) : 0.f;
  }
  return MH_OK;
}

mh_status mh_map_get_info(const mh_map* m, mh_map_info* info) {
  MH_REQUIRE(m && info, "null argument");
  info->n_points = m->n_points;
  info->n_offered = m->n_offered;
  info->n_voxels = m->n_voxels;
  info->table_size = m->table_size;
  for (int a = 0; a < 3; a++) {
    info->bbox_min[a] = m->bbox_min[a];
    info->bbox_max[a] = m->bbox_max[a];
  }
  info->voxel_size = m->params.voxel_size;
  info->max_points_per_voxel = m->params.max_points_per_voxel;
  return MH_OK;
}

mh_status mh_map_download(const mh_map* m, float* x, float* y, float* z, uint32_t* src_idx, int32_t* vox_keys_xyz,
                          uint32_t* vox_first, uint32_t* vox_count) {
  MH_REQUIRE(m, "null map");
  mh_ctx* ctx = m->ctx;
  MH_TRY(set_device(ctx));
  MH_HIP(hipStreamSynchronize(ctx->stream));
  if (m->n_points && (x || y || z || src_idx)) {
    std::vector<float4> h(m->n_points);
    MH_HIP(hipMemcpy(h.data(), m->pts.p, m->n_points * sizeof(float4), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < m->n_points; i++) {
      if (x) x[i] = h[i].x;
      if (y) y[i] = h[i].y;
      if (z) z[i] = h[i].z;
      if (src_idx) memcpy(&src_idx[i], &h[i].w, 4);
    }
  }
  if (m->n_voxels) {
    if (vox_keys_xyz) {
      std::vector<unsigned long long> k(m->n_voxels);
      MH_HIP(hipMemcpy(k.data(), m->vox_keys.p, m->n_voxels * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      for (size_t v = 0; v < m->n_voxels; v++) {
        int kx, ky, kz;
        unpack_key(k[v], kx, ky, kz);
        vox_keys_xyz[3 * v] = kx;
        vox_keys_xyz[3 * v + 1] = ky;
        vox_keys_xyz[3 * v + 2] = kz;
      }
    }
    if (vox_first) MH_HIP(hipMemcpy(vox_first, m->vox_first.p, m->n_voxels * 4, hipMemcpyDeviceToHost));
    if (vox_count) MH_HIP(hipMemcpy(vox_count, m->vox_count.p, m->n_voxels * 4, hipMemcpyDeviceToHost));
  }
  return MH_OK;
}

}  // extern "C"
