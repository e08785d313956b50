// mh_dev_variants.h -- the matcher families that were built, measured against the product kernels and lost:
//   MH_MATCH=t  k_match_tile    north_star's kernel: a workgroup per tile of the spatially sorted scan, map records staged in LDS
//   MH_MATCH=w  k_match_wave_*  a wave per tile of 64 sorted points, wave-uniform candidates through the scalar path / LDS
//   MH_MATCH=o  k_match4o_b     the quad matcher over the scan in search order
// Bit-identical pairings, parity-tested, slower (profiles/r02_tile_matcher.md).  Compiled only with -DMH_DEV_VARIANTS
// (tools/build_variants.sh -> tools/variants/libmolahip_dev.so, with mh_tile.hip); the shipped library does not carry them.
// Included by mh_icp.hip after the product matchers' kernels.
#pragma once

// ================================================================================================
// k_match_tile: correspondence search of a large layer with the map records staged in LDS, one workgroup per tile of the
// spatially sorted scan (nn_search_tile, mh_tile.hip).  Pairings are written at the points' ORIGINAL indices, so
// everything downstream (k_accum, covariance, compaction of the final pairings) is what it is for the other matchers.
// ================================================================================================
__device__ __forceinline__ void k_match_tile_body(const IcpDeviceState* __restrict__ st, const float* __restrict__ sx,
                                                  const float* __restrict__ sy, const float* __restrict__ sz,
                                                  const uint32_t* __restrict__ perm, const uint32_t* __restrict__ tile_start,
                                                  uint32_t n_tiles, MapView map, float4* __restrict__ pair_q,
                                                  uint32_t* __restrict__ pair_gidx
#ifdef MH_DEBUG_WAVETRACE
                                                  , unsigned long long* __restrict__ wtrace
#endif
) {
  __shared__ TileShared sh;
  const uint32_t t = blockIdx.x;
  if (t >= n_tiles) return;
#ifdef MH_DEBUG_WAVETRACE
  unsigned long long* dbg = wtrace ? wtrace + 8ull * t : nullptr;  // [start, bbox, probe, copy, search, end, nvox, records]
  if (threadIdx.x == 0 && dbg) dbg[0] = wall_clock64();
#endif
  const uint32_t s0 = tile_start[t], s1 = tile_start[t + 1];
  const uint32_t i = s0 + threadIdx.x;
  const bool active = i < s1;
  const uint32_t ic = active ? i : s0;
  const float x = sx[ic], y = sy[ic], z = sz[ic];
  const uint32_t orig = perm[ic];
  const uint32_t done = st->done;
  double T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = st->T[k];
  const float thr2 = st->cur_thr2, ang2 = st->cur_ang2;
  if (done) return;  // grid-uniform
  float px, py, pz;
  transform_point(T, x, y, z, px, py, pz);
  const NNResult r = nn_search_tile(map, sh, active, px, py, pz
#ifdef MH_DEBUG_WAVETRACE
                                    , dbg
#endif
  );
  if (active) {
    const float n2 = (px * px + py * py) + pz * pz;
    const bool ok = r.found && (r.d2 < thr2 + ang2 * n2);
    pair_q[orig] = make_float4(r.pt.x, r.pt.y, r.pt.z, r.d2);
    pair_gidx[orig] = ok ? __float_as_uint(r.pt.w) : kNoMatch;
  }
#ifdef MH_DEBUG_WAVETRACE
  if (threadIdx.x == 0 && dbg) dbg[5] = wall_clock64();
#endif
}


// the quad matcher over the scan in search order (mh_tile.hip): neighbouring quads need the same voxels and tend to take the
// same number of rounds
__global__ __launch_bounds__(kBlock, MH_QUAD_WAVES) void k_match4o_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  k_match4_body(j.st, j.sx, j.sy, j.sz, j.n, j.map, j.pair_q, j.pair_gidx, j.perm
#ifdef MH_DEBUG_WAVETRACE
                , nullptr
#endif
  );
}
__global__ __launch_bounds__(kTileThreads) void k_match_tile(const IcpDeviceState* __restrict__ st, const float* __restrict__ sx,
                                                             const float* __restrict__ sy, const float* __restrict__ sz,
                                                             const uint32_t* __restrict__ perm,
                                                             const uint32_t* __restrict__ tile_start, uint32_t n_tiles,
                                                             MapView map, float4* __restrict__ pair_q,
                                                             uint32_t* __restrict__ pair_gidx
#ifdef MH_DEBUG_WAVETRACE
                                                             , unsigned long long* __restrict__ wtrace
#endif
) {
  k_match_tile_body(st, sx, sy, sz, perm, tile_start, n_tiles, map, pair_q, pair_gidx
#ifdef MH_DEBUG_WAVETRACE
                    , wtrace
#endif
  );
}
__global__ __launch_bounds__(kTileThreads) void k_match_tile_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  k_match_tile_body(j.st, j.sx, j.sy, j.sz, j.perm, j.tile_start, j.n_tiles, j.map, j.pair_q, j.pair_gidx
#ifdef MH_DEBUG_WAVETRACE
                    , nullptr
#endif
  );
}
// k_match_wave: tiles of <= 64 spatially sorted points.  Two launches per iteration over the same tile table:
//   DENSE   one wave (a 64-thread workgroup) per tile with >= kWaveMinPoints points: wave-uniform candidates
//           (nn_search_wave: scalar loads; LDS: the box's records staged in LDS instead when they fit);
//   sparse  the other tiles by quads (nn_search_quad), sixteen points per wave, four waves per tile.
// Each tile is handled by exactly one of the two; the other launch's workgroup leaves at once.  Two kernels instead of
// one so that each gets the registers of its own path only.
template <bool DENSE, bool LDS>
__device__ __forceinline__ void k_match_wave_body(const IcpDeviceState* __restrict__ st, const float* __restrict__ sx,
                                                  const float* __restrict__ sy, const float* __restrict__ sz,
                                                  const uint32_t* __restrict__ perm, const uint32_t* __restrict__ tile_start,
                                                  uint32_t n_tiles, MapView map, float4* __restrict__ pair_q,
                                                  uint32_t* __restrict__ pair_gidx
#ifdef MH_DEBUG_WAVETRACE
                                                  , unsigned long long* __restrict__ wtrace
#endif
) {
  WaveShared* wsh = nullptr;
  if constexpr (LDS && DENSE) {
    __shared__ WaveShared wsh_store;
    wsh = &wsh_store;
  }
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t t = blockIdx.x;
  if (t >= n_tiles) return;
  // everything wave-uniform comes in through the scalar path (constant address space, uniform addresses): the pose lives
  // in SGPRs, not in 24 VGPRs per lane
  typedef const uint32_t __attribute__((address_space(4))) * cu32_ptr;
  typedef const IcpDeviceState __attribute__((address_space(4))) * cstate_ptr;
  const cu32_ptr cts = (cu32_ptr)uniform_const_ptr(tile_start);
  const cstate_ptr cst = (cstate_ptr)uniform_const_ptr(st);
  const uint32_t s0 = cts[t], s1 = cts[t + 1];
  const uint32_t cnt = s1 - s0;
  if ((cnt >= kWaveMinPoints) != DENSE) return;  // the other launch's tile
  const uint32_t done = cst->done;
  double T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = cst->T[k];
  const float thr2 = cst->cur_thr2, ang2 = cst->cur_ang2;
  if (done) return;  // grid-uniform
  if (DENSE) {
    const uint32_t i = s0 + lane;
    const bool active = i < s1;
    const uint32_t ic = active ? i : s0;
    const float x = sx[ic], y = sy[ic], z = sz[ic];
    const uint32_t orig = perm[ic];
    float px, py, pz;
    transform_point(T, x, y, z, px, py, pz);
#ifdef MH_DEBUG_WAVETRACE
    unsigned long long* dbg = wtrace ? wtrace + 8ull * t : nullptr;  // [start, end, points, voxels, probed, copied, pass 1, records]
    if (lane == 0 && dbg) { dbg[0] = wall_clock64(); dbg[2] = cnt; }
#endif
    const NNResult r = nn_search_wave<LDS>(map, wsh, active, px, py, pz
#ifdef MH_DEBUG_WAVETRACE
                                           , dbg
#endif
    );
#ifdef MH_DEBUG_WAVETRACE
    if (lane == 0 && dbg) dbg[1] = wall_clock64();
#endif
    if (active) {
      const float n2 = (px * px + py * py) + pz * pz;
      const bool ok = r.found && (r.d2 < thr2 + ang2 * n2);
      pair_q[orig] = make_float4(r.pt.x, r.pt.y, r.pt.z, r.d2);
      pair_gidx[orig] = ok ? __float_as_uint(r.pt.w) : kNoMatch;
    }
    return;
  }
  const uint32_t sub = lane & 3u;
  {
    const uint32_t base = 16u * wave;  // sixteen points per wave, a quad each
    if (base >= cnt) return;
    const uint32_t q = base + (lane >> 2);
    const bool active = q < cnt;
    const uint32_t ic = s0 + (active ? q : 0u);
    const float x = sx[ic], y = sy[ic], z = sz[ic];
    const uint32_t orig = perm[ic];
    float px, py, pz;
    transform_point(T, x, y, z, px, py, pz);
#ifdef MH_DEBUG_WAVETRACE
    unsigned long long* dbg = (wtrace && wave == 0) ? wtrace + 8ull * t : nullptr;
    if (lane == 0 && dbg) { dbg[0] = wall_clock64(); dbg[2] = cnt; dbg[3] = 0; }
#endif
    const NNResult r = nn_search_quad(map, sub, px, py, pz);
    if (active && sub == 0u) {
      const float n2 = (px * px + py * py) + pz * pz;
      const bool ok = r.found && (r.d2 < thr2 + ang2 * n2);
      pair_q[orig] = make_float4(r.pt.x, r.pt.y, r.pt.z, r.d2);
      pair_gidx[orig] = ok ? __float_as_uint(r.pt.w) : kNoMatch;
    }
#ifdef MH_DEBUG_WAVETRACE
    if (lane == 0 && dbg) dbg[1] = wall_clock64();
#endif
  }
}
#ifdef MH_DEBUG_WAVETRACE
#define MH_WT_PARAM , unsigned long long* __restrict__ wtrace
#define MH_WT_ARG , wtrace
#define MH_WT_NULL , nullptr
#define MH_WT_G , g_wtrace
#else
#define MH_WT_PARAM
#define MH_WT_ARG
#define MH_WT_NULL
#endif
template <bool LDS>
__global__ __launch_bounds__(64, 8) void k_match_wave_dense(const IcpDeviceState* __restrict__ st, const float* __restrict__ sx,
                                                            const float* __restrict__ sy, const float* __restrict__ sz,
                                                            const uint32_t* __restrict__ perm,
                                                            const uint32_t* __restrict__ tile_start, uint32_t n_tiles, MapView map,
                                                            float4* __restrict__ pair_q, uint32_t* __restrict__ pair_gidx MH_WT_PARAM) {
  k_match_wave_body<true, LDS>(st, sx, sy, sz, perm, tile_start, n_tiles, map, pair_q, pair_gidx MH_WT_ARG);
}
__global__ __launch_bounds__(kBlock, MH_QUAD_WAVES) void k_match_wave_sparse(const IcpDeviceState* __restrict__ st,
                                                                            const float* __restrict__ sx, const float* __restrict__ sy,
                                                                            const float* __restrict__ sz, const uint32_t* __restrict__ perm,
                                                                            const uint32_t* __restrict__ tile_start, uint32_t n_tiles,
                                                                            MapView map, float4* __restrict__ pair_q,
                                                                            uint32_t* __restrict__ pair_gidx MH_WT_PARAM) {
  k_match_wave_body<false, false>(st, sx, sy, sz, perm, tile_start, n_tiles, map, pair_q, pair_gidx MH_WT_ARG);
}
template <bool LDS>
__global__ __launch_bounds__(64, 8) void k_match_wave_dense_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  k_match_wave_body<true, LDS>(j.st, j.sx, j.sy, j.sz, j.perm, j.tile_start, j.n_tiles, j.map, j.pair_q, j.pair_gidx MH_WT_NULL);
}
__global__ __launch_bounds__(kBlock, MH_QUAD_WAVES) void k_match_wave_sparse_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  k_match_wave_body<false, false>(j.st, j.sx, j.sy, j.sz, j.perm, j.tile_start, j.n_tiles, j.map, j.pair_q, j.pair_gidx MH_WT_NULL);
}
// both launches of the wave matcher on stream s (LDS staging of the dense tiles: MH_WAVE_LDS=1)
static inline bool wave_lds_env() { static const bool v = getenv("MH_WAVE_LDS") != nullptr; return v; }
#define MH_LAUNCH_WAVE(S, ST, SC, MV, PQ, PG, WT)                                                                          \
  do {                                                                                                                     \
    if (wave_lds_env())                                                                                                    \
      hipLaunchKernelGGL(k_match_wave_dense<true>, dim3((SC)->n_tiles), dim3(64), 0, S, ST, (SC)->sx, (SC)->sy, (SC)->sz,   \
                         (SC)->perm, (SC)->tile_start, (SC)->n_tiles, MV, PQ, PG WT);                                      \
    else                                                                                                                   \
      hipLaunchKernelGGL(k_match_wave_dense<false>, dim3((SC)->n_tiles), dim3(64), 0, S, ST, (SC)->sx, (SC)->sy, (SC)->sz,  \
                         (SC)->perm, (SC)->tile_start, (SC)->n_tiles, MV, PQ, PG WT);                                      \
    hipLaunchKernelGGL(k_match_wave_sparse, dim3((SC)->n_tiles), dim3(kBlock), 0, S, ST, (SC)->sx, (SC)->sy, (SC)->sz,      \
                       (SC)->perm, (SC)->tile_start, (SC)->n_tiles, MV, PQ, PG WT);                                        \
  } while (0)
