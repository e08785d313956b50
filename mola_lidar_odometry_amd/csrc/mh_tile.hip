// mh_tile.hip -- search order of the tile matcher: a scan's points sorted by where they are, cut into tiles.
//
// The correspondence search of a large layer (Matcher_Points_DistanceThreshold [U], lidar3d-default.yaml:195-204, on the
// 120 k-point layers of BASELINE configs[1]) is a gather: every point needs the map records of the 27 voxels around it.
// Points that are close to each other need the SAME records, and a LiDAR scan is dense where it matters (C2: 120 k points
// fall into 2.5 k voxels).  So the points are sorted once per scan -- in the LOCAL frame, because a rigid transform keeps
// neighbours together whatever the pose of the iteration is -- by
//     (2x2x2-voxel block, quarter-voxel Morton code inside the block)
// and cut into tiles of at most `tile_points` (256: k_match_tile, 64: k_match_wave) consecutive points that never cross a block boundary.  One workgroup of
// k_match_tile (mh_icp.hip) then loads the union of its tile's neighbourhoods into LDS once per iteration (the box around
// the transformed points, at most a few hundred records) and every point searches LDS.
//
// Everything here runs on the scan's stream, asynchronously; the only thing the host needs is the tile count.
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "mh_internal.h"

namespace mh {

constexpr unsigned long long kTileKeyInvalid = (1ull << 39) - 1;  // non-finite / far-away points sort last

__device__ __forceinline__ uint32_t spread3(uint32_t v) {  // 3 bits -> every third bit
  return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4);
}

__global__ __launch_bounds__(256) void k_tile_keys(const float* __restrict__ x, const float* __restrict__ y,
                                                   const float* __restrict__ z, uint32_t n, float inv_q,
                                                   unsigned long long* __restrict__ keys, uint32_t* __restrict__ idx) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float fx = floorf(x[i] * inv_q), fy = floorf(y[i] * inv_q), fz = floorf(z[i] * inv_q);  // quarter-voxel cell
  unsigned long long key = kTileKeyInvalid;
  const float lim = 4000.f;  // |block index| < 512
  if (fabsf(fx) < lim && fabsf(fy) < lim && fabsf(fz) < lim) {  // (NaN / inf fail the test)
    const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
    const uint32_t bx = (uint32_t)((cx >> 3) + 512), by = (uint32_t)((cy >> 3) + 512), bz = (uint32_t)((cz >> 3) + 512);
    const uint32_t fine = (spread3((uint32_t)cx & 7u) << 2) | (spread3((uint32_t)cy & 7u) << 1) | spread3((uint32_t)cz & 7u);
    key = ((unsigned long long)((bx << 20) | (by << 10) | bz) << 9) | fine;
  }
  keys[i] = key;
  idx[i] = i;
}

// sorted copy of the coordinates + "a tile starts here" flags: a new block, or every tile_points-th sorted position
__global__ __launch_bounds__(256) void k_tile_heads(const unsigned long long* __restrict__ keys_s,
                                                    const uint32_t* __restrict__ perm, const float* __restrict__ x,
                                                    const float* __restrict__ y, const float* __restrict__ z, uint32_t n,
                                                    float* __restrict__ sx, float* __restrict__ sy, float* __restrict__ sz,
                                                    uint8_t* __restrict__ head, uint32_t tile_points) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t p = perm[i];
  sx[i] = x[p];
  sy[i] = y[p];
  sz[i] = z[p];
  const unsigned long long b = keys_s[i] >> 9;
  head[i] = (i == 0 || (i % tile_points) == 0 || (keys_s[i - 1] >> 9) != b) ? 1 : 0;
}

__global__ void k_tile_finish(uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ count, uint32_t n,
                              uint32_t* __restrict__ h_count) {
  const uint32_t c = *count;
  tile_start[c] = n;  // end of the last tile
  *h_count = c;       // pinned host memory
}

uint32_t tile_points_for_env() {  // MH_MATCH=w (wave matcher): tiles of one wave; t: tiles of one workgroup
  const char* e = getenv("MH_MATCH");
  return (e && e[0] == 'w') ? 64u : 256u;
}

void scan_drop_tiles(mh_scan* s) {
  s->tiles_valid = false;
  s->tiles_pending = false;
  s->n_tiles = 0;
}

void scan_free_tiles(mh_scan* s) {
  scan_drop_tiles(s);
  s->tiles.release();
  if (s->h_ntiles) (void)hipHostFree(s->h_ntiles);
  if (s->ev_tiles) (void)hipEventDestroy(s->ev_tiles);
  s->h_ntiles = nullptr;
  s->ev_tiles = nullptr;
}

mh_status scan_build_tiles(const mh_scan* s, float inv_vs, uint32_t tile_points) {
  if (s->tiles_valid && s->tile_inv_vs == inv_vs && s->tile_points == tile_points) return MH_OK;
  mh_ctx* ctx = s->ctx;
  MH_TRY(set_device(ctx));
  hipStream_t st = ctx->stream;
  const size_t n = s->n;
  const uint32_t N = (uint32_t)n;
  if (!s->h_ntiles) {
    MH_HIP(hipHostMalloc((void**)&s->h_ntiles, 64, hipHostMallocDefault));
    MH_HIP(hipEventCreateWithFlags(&s->ev_tiles, hipEventDisableTiming));
  }
  const size_t stride = ((n * sizeof(float) + 255) / 256) * 256;
  const size_t need = 4 * stride + ((n + 2) * sizeof(uint32_t) + 255) / 256 * 256;
  if (s->tiles.bytes < need) {
    MH_HIP(mh::wait_stream(st));  // nobody may still read the old buffer
    MH_TRY(s->tiles.reserve(need));
  }
  char* base = s->tiles.as<char>();
  float *sx = (float*)base, *sy = (float*)(base + stride), *sz = (float*)(base + 2 * stride);
  uint32_t* perm = (uint32_t*)(base + 3 * stride);
  uint32_t* tile_start = (uint32_t*)(base + 4 * stride);
  s->sx = sx; s->sy = sy; s->sz = sz; s->perm = perm; s->tile_start = tile_start;
  s->tiles_valid = false;  // (set again only after the last enqueue below has succeeded: a failed build must not look valid)
  s->tile_inv_vs = inv_vs;
  s->tile_points = tile_points;
  if (n == 0) {
    s->n_tiles = 0;
    s->tiles_pending = false;
    s->tiles_valid = true;
    return MH_OK;
  }
  // scratch: keys | sorted keys | indices | flags | count, in the context's build buffers
  MH_TRY(ctx->build_a.reserve(2 * n * sizeof(unsigned long long)));
  MH_TRY(ctx->build_b.reserve(n * sizeof(uint32_t)));
  MH_TRY(ctx->build_c.reserve(n + 256));
  unsigned long long* keys = ctx->build_a.as<unsigned long long>();
  unsigned long long* keys_s = keys + n;
  uint32_t* idx = ctx->build_b.as<uint32_t>();
  uint8_t* head = ctx->build_c.as<uint8_t>();
  uint32_t* count = reinterpret_cast<uint32_t*>(head + (n + 63) / 64 * 64);
  size_t tmp = 0, t2 = 0;
  MH_HIP(rocprim::radix_sort_pairs(nullptr, tmp, keys, keys_s, idx, perm, N, 0, 39, st));
  MH_HIP(rocprim::select(nullptr, t2, rocprim::counting_iterator<uint32_t>(0), head, tile_start, count, N, st));
  if (t2 > tmp) tmp = t2;
  MH_TRY(ctx->sort_tmp.reserve(tmp));
  const uint32_t nb = (N + 255) / 256;
  hipLaunchKernelGGL(k_tile_keys, dim3(nb), dim3(256), 0, st, s->x, s->y, s->z, N, inv_vs * 4.0f, keys, idx);
  size_t tb = ctx->sort_tmp.bytes;
  MH_HIP(rocprim::radix_sort_pairs(ctx->sort_tmp.p, tb, keys, keys_s, idx, perm, N, 0, 39, st));
  hipLaunchKernelGGL(k_tile_heads, dim3(nb), dim3(256), 0, st, keys_s, perm, s->x, s->y, s->z, N, sx, sy, sz, head, tile_points);
  tb = ctx->sort_tmp.bytes;
  MH_HIP(rocprim::select(ctx->sort_tmp.p, tb, rocprim::counting_iterator<uint32_t>(0), head, tile_start, count, N, st));
  hipLaunchKernelGGL(k_tile_finish, dim3(1), dim3(1), 0, st, tile_start, count, N, s->h_ntiles);
  MH_HIP(hipGetLastError());
  MH_HIP(hipEventRecord(s->ev_tiles, st));
  s->tiles_pending = true;
  s->tiles_valid = true;
  return MH_OK;
}

mh_status scan_tiles_ready(const mh_scan* s) {
  if (!s->tiles_valid) return fail(MH_ERR_INTERNAL, "scan_tiles_ready without scan_build_tiles");
  if (s->tiles_pending) {
    MH_TRY(set_device(s->ctx));
    MH_HIP(mh::wait_event(s->ev_tiles));
    s->n_tiles = *s->h_ntiles;
    s->tiles_pending = false;
  }
  return MH_OK;
}

}  // namespace mh

extern "C" {
// The tile order is built by the first alignment that wants it; a caller that uploads scans ahead of time (bench.py, a
// replay that prefetches) can queue the build right behind the upload so that it overlaps whatever runs meanwhile.
mh_status mh_scan_prepare(const mh_scan* scan, float voxel_size) {
  MH_REQUIRE(scan, "null scan");
  MH_REQUIRE(voxel_size > 0.f, "voxel_size must be > 0");
  return mh::scan_build_tiles(scan, 1.0f / voxel_size, mh::tile_points_for_env());
}
}
