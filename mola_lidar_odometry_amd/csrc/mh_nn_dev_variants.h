// mh_nn_dev_variants.h -- the searches of the development matcher families (mh_dev_variants.h): nn_search_tile (map records of a
// tile's box staged in LDS) and nn_search_wave (wave-uniform candidates).  Included by mh_nn_device.h under -DMH_DEV_VARIANTS only.
#pragma once

// -------------------------------------------------------------------------------------------------
// Tile search: one WORKGROUP per tile of <= 256 spatially sorted scan points (mh_tile.hip), one lane per point.
//   1. the workgroup takes the box of voxels around its transformed points (their bounding box + 1 voxel each way;
//      the points of a tile come from one 2x2x2-voxel block of the local frame, so the box has <= ~150 voxels);
//   2. thread v probes box voxel v (ONE round trip for the whole box), an exclusive scan of the counts in box order
//      assigns every voxel its place in LDS;
//   3. the records of the occupied voxels are copied to LDS, a DPP row (16 lanes) per voxel: ONE more round trip,
//      coalesced;  C2: 3.8 records loaded per scan point, where the per-point searches read ~35 candidates + ~8 slots;
//   4. every lane scans its 27 voxels in LDS: own voxel, then faces, edges, corners, a voxel being skipped by the whole
//      wave when no lane's bound admits it (the 64 points of a wave are neighbours, they want the same voxels).
// Box order (x outer, y, z inner) is ascending packed-key order, i.e. ascending record index, so the position in LDS
// orders candidates exactly as the record index does: the key (d2 bits << 32 | LDS position) has the reference's
// tie-break order.  Same candidates (27 voxels), same fp32 arithmetic, same strict minimum: bit-identical pairings.
// A tile whose box or record count does not fit (never on C2) falls back to nn_search_pruned, lane by lane.
// -------------------------------------------------------------------------------------------------
constexpr int kTileMaxVox = 256;    // box voxels (one probe per thread)
constexpr int kTileMaxRec = 1152;   // records in LDS (18 KiB); C2: median 280, maximum 1076
constexpr int kTileThreads = 256;

struct TileShared {
  f32x4 rec[kTileMaxRec];
  uint32_t vt[kTileMaxVox];       // LDS position of the voxel's first record | count << 16
  uint32_t vfirst[kTileMaxVox];   // record index of the voxel's first record in the map
  uint32_t occ[kTileMaxVox];      // occupied voxels, compacted
  int red[6][4];                  // per-wave bounding box
  uint32_t wtot[4], wocc[4];
};

__device__ __forceinline__ void tile_scan_run(const TileShared& sh, uint32_t off, uint32_t cnt, float qx, float qy, float qz,
                                              nnkey_t& best) {
  for (uint32_t j = 0; __ballot(j < cnt) != 0ull; j += 4u) {
    f32x4 c[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      ok[u] = j + (uint32_t)u < cnt;
      c[u] = sh.rec[ok[u] ? off + j + (uint32_t)u : 0u];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const float dx = c[u].x - qx, dy = c[u].y - qy, dz = c[u].z - qz;
      const float d2 = (dx * dx + dy * dy) + dz * dz;  // fp32, un-fused, this order (bit-exact with the oracle)
      const nnkey_t k = ok[u] ? (((nnkey_t)__float_as_uint(d2) << 32) | (off + j + (uint32_t)u)) : kNNKeyNone;
      best = k < best ? k : best;
    }
  }
}

// The whole workgroup calls this (it contains barriers).  `active`: the lane holds a point of the tile.  Returns the
// nearest map point of the lane's query (every lane its own).
#ifdef MH_DEBUG_WAVETRACE
#define MH_WDBG(...) do { if ((threadIdx.x & 63u) == 0 && dbg) { __VA_ARGS__; } } while (0)
#define MH_TILE_DBG_ARG , unsigned long long* __restrict__ dbg
#define MH_TSTAMP(i) do { if (threadIdx.x == 0 && dbg) dbg[i] = wall_clock64(); } while (0)
#define MH_TVALUE(i, v) do { if (threadIdx.x == 0 && dbg) dbg[i] = (unsigned long long)(v); } while (0)
#else
#define MH_WDBG(...) do { } while (0)
#define MH_TILE_DBG_ARG
#define MH_TSTAMP(i) do { } while (0)
#define MH_TVALUE(i, v) do { } while (0)
#endif
__device__ __forceinline__ NNResult nn_search_tile(const MapView& m, TileShared& sh, bool active, float qx, float qy, float qz MH_TILE_DBG_ARG) {
  NNResult r;
  r.d2 = __builtin_inff();
  r.found = false;
  r.pt = (f32x4)(0.f);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const float lim = 1.0e6f;  // one test, no short-circuit branches: NaN and inf fail it as well
  const bool valid = active && ((int)(fabsf(qx * m.inv_vs) < lim) & (int)(fabsf(qy * m.inv_vs) < lim) & (int)(fabsf(qz * m.inv_vs) < lim));
  const int cx = valid ? voxel_of(qx, m.inv_vs, m.trunc) : 0, cy = valid ? voxel_of(qy, m.inv_vs, m.trunc) : 0,
            cz = valid ? voxel_of(qz, m.inv_vs, m.trunc) : 0;
  {  // bounding box of the tile's voxels
    const int big = 0x40000000;
    const int lo_x = wave_min_i32(valid ? cx : big), lo_y = wave_min_i32(valid ? cy : big), lo_z = wave_min_i32(valid ? cz : big);
    const int hi_x = wave_max_i32(valid ? cx : -big), hi_y = wave_max_i32(valid ? cy : -big), hi_z = wave_max_i32(valid ? cz : -big);
    if (lane == 0) {
      sh.red[0][wave] = lo_x; sh.red[1][wave] = lo_y; sh.red[2][wave] = lo_z;
      sh.red[3][wave] = hi_x; sh.red[4][wave] = hi_y; sh.red[5][wave] = hi_z;
    }
  }
  __syncthreads();
  int lo[3], hi[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    lo[a] = min(min(sh.red[a][0], sh.red[a][1]), min(sh.red[a][2], sh.red[a][3]));
    hi[a] = max(max(sh.red[a + 3][0], sh.red[a + 3][1]), max(sh.red[a + 3][2], sh.red[a + 3][3]));
  }
  MH_TSTAMP(1);
  if (hi[0] < lo[0]) return r;  // no valid point in the tile (workgroup-uniform)
  const int ox = lo[0] - 1, oy = lo[1] - 1, oz = lo[2] - 1;
  const long long ex = (long long)hi[0] - lo[0] + 3, ey = (long long)hi[1] - lo[1] + 3, ez = (long long)hi[2] - lo[2] + 3;
  const long long nvox_l = ex * ey * ez;
  const gslots_ptr slots4 = (gslots_ptr)m.slots;
  const gpts_ptr pts4 = (gpts_ptr)m.pts;
  bool fits = nvox_l <= (long long)kTileMaxVox;
  const int dy = (int)ey, dz = (int)ez, dydz = dy * dz;
  const uint32_t nvox = fits ? (uint32_t)nvox_l : 0u;
  uint32_t first = 0, cnt = 0;
  if (fits && tid < nvox) {  // thread v <-> box voxel v = (ix * dy + iy) * dz + iz
    const uint32_t t = (uint32_t)(((float)tid + 0.5f) * (1.0f / (float)dz));  // exact for these ranges (margin 1/(2 dz))
    const uint32_t iz = tid - t * (uint32_t)dz;
    const uint32_t ix = (uint32_t)(((float)t + 0.5f) * (1.0f / (float)dy));
    const uint32_t iy = t - ix * (uint32_t)dy;
    const unsigned long long key = pack_key(ox + (int)ix, oy + (int)iy, oz + (int)iz);
    nn_resolve(m, slots4, key, slots4[hash_key(key) & m.mask], true, first, cnt);
  }
  const uint32_t incl = wave_scan_incl(cnt);
  const unsigned long long occ_mask = __ballot(cnt > 0u);
  if (lane == 63) sh.wtot[wave] = incl;
  if (lane == 0) sh.wocc[wave] = (uint32_t)__popcll(occ_mask);
  __syncthreads();
  uint32_t base = 0, obase = 0, total = 0, n_occ = 0;
#pragma unroll
  for (uint32_t w = 0; w < 4; w++) {
    base += w < wave ? sh.wtot[w] : 0u;
    obase += w < wave ? sh.wocc[w] : 0u;
    total += sh.wtot[w];
    n_occ += sh.wocc[w];
  }
  fits = fits && total <= (uint32_t)kTileMaxRec;
  MH_TSTAMP(2);
  MH_TVALUE(6, nvox_l);
  MH_TVALUE(7, total | ((unsigned long long)n_occ << 32));
  if (!fits) {  // workgroup-uniform: the lane-by-lane search through the caches
    return valid ? nn_search_pruned(m, qx, qy, qz) : r;
  }
  if (tid < nvox) {
    const uint32_t off = base + incl - cnt;
    sh.vt[tid] = off | (cnt << 16);
    sh.vfirst[tid] = first;
    if (cnt > 0u) sh.occ[obase + (uint32_t)__popcll(occ_mask & ((1ull << lane) - 1ull))] = tid;
  }
  __syncthreads();
  {  // records -> LDS, sixteen lanes per occupied voxel
    const uint32_t r16 = tid & 15u;
    for (uint32_t k = tid >> 4; k < n_occ; k += kTileThreads / 16) {
      const uint32_t v = sh.occ[k];
      const uint32_t oc = sh.vt[v], gf = sh.vfirst[v];
      const uint32_t off = oc & 0xFFFFu, c = oc >> 16;
      for (uint32_t j = r16; j < c; j += 16u) sh.rec[off + j] = pts4[gf + j];
    }
  }
  __syncthreads();
  MH_TSTAMP(3);
  const Gaps gx = axis_gaps(qx, cx, m.vs, m.trunc), gy = axis_gaps(qy, cy, m.vs, m.trunc), gz = axis_gaps(qz, cz, m.vs, m.trunc);
  const int b0 = ((cx - ox) * dy + (cy - oy)) * dz + (cz - oz);
  nnkey_t best = kNNKeyNone;
  {  // the query's own voxel
    const uint32_t oc = valid ? sh.vt[b0] : 0u;
    tile_scan_run(sh, oc & 0xFFFFu, oc >> 16, qx, qy, qz, best);
  }
  // the neighbours that can still hold a candidate with d2 <= best: bit `code` of `mask` (code = ix * 9 + iy * 3 + iz)
  uint32_t mask = 0;
  if (valid) {
#pragma unroll
    for (int c = 0; c < 27; c++) {
      if (c == 13) continue;
      const float lb = (gx.s[c / 9] + gy.s[(c / 3) % 3]) + gz.s[c % 3];
      if (!(lb * 0.9999f > nnkey_d2(best))) mask |= 1u << c;
    }
  }
  // every lane walks ITS OWN live voxels (faces, then edges, then corners: nearer voxels tighten the bound for the
  // farther ones); the wave stays in the loop as long as one lane has a voxel left
  const uint32_t kFaces = (1u << 4) | (1u << 10) | (1u << 12) | (1u << 14) | (1u << 16) | (1u << 22);
  const uint32_t kCorners = (1u << 0) | (1u << 2) | (1u << 6) | (1u << 8) | (1u << 18) | (1u << 20) | (1u << 24) | (1u << 26);
  const uint32_t kEdges = 0x07FFFFFFu & ~(kFaces | kCorners | (1u << 13));
#pragma unroll 1
  for (int cls = 0; cls < 3; cls++) {
    uint32_t mm = mask & (cls == 0 ? kFaces : (cls == 1 ? kEdges : kCorners));
    while (__ballot(mm != 0u) != 0ull) {
      uint32_t off = 0, cnt = 0;
      while (mm != 0u && cnt == 0u) {  // next live voxel of this lane that holds records and still passes the bound
        const int c = __builtin_ctz(mm);
        mm &= mm - 1u;
        if (nn_lower_bound(c, gx, gy, gz) * 0.9999f > nnkey_d2(best)) continue;
        const int ix = (c * 57) >> 9, rr = c - 9 * ix, iy = (rr * 11) >> 5, iz = rr - 3 * iy;
        const uint32_t oc = sh.vt[b0 + (ix - 1) * dydz + (iy - 1) * dz + (iz - 1)];
        off = oc & 0xFFFFu;
        cnt = oc >> 16;
      }
      tile_scan_run(sh, off, cnt, qx, qy, qz, best);
    }
  }
  MH_TSTAMP(4);
  if (valid && nnkey_idx(best) != 0xFFFFFFFFu) {
    r.pt = sh.rec[nnkey_idx(best)];
    r.d2 = nnkey_d2(best);
    r.found = true;
  }
  return r;
}

// -------------------------------------------------------------------------------------------------
// Wave search: one WAVE per tile of <= 64 spatially sorted scan points (mh_tile.hip with tile_points = 64), one lane per
// point, and the CANDIDATES ARE WAVE-UNIFORM.  The 64 points of a tile are neighbours (C2: 88 % of the points sit in
// voxels that hold >= 32 of them), so they want the same map records; instead of every lane (or quad) fetching its own
// copy through the vector path,
//   1. the wave takes the box of voxels around its transformed points (bounding box + 1 each way, <= 128 voxels), lane v
//      probes box voxel v (and v + 64): one round trip;
//   2. pass 1: for every distinct own voxel of the wave, the lanes that live in it take the minimum d2 over its records
//      -- a bound, nothing else;
//   3. pass 2: the occupied box voxels in ascending order (= ascending record index); a voxel is scanned when at least
//      one lane has it inside its 27-voxel block AND cannot rule it out by its bound; its records are read through the
//      SCALAR path (the address is wave-uniform: s_load_dwordx4, no vector memory instruction, no LDS, no barrier) and
//      every interested lane tests the same record against its own query: ~11 vector instructions per candidate per WAVE.
//      Ascending order makes a strict '<' the reference's "first minimum in scan order".
// Tiles with fewer than kWaveMinPoints points are searched by the wave's sixteen quads instead (nn_search_quad), sixteen
// points per pass: for scattered points the wave-uniform loop would scan every voxel near ANY of them for all of them.
// -------------------------------------------------------------------------------------------------
// LDS hand-off between lanes of ONE wave: its LDS operations execute in order, only the compiler has to be kept from
// moving the accesses across this point
constexpr uint32_t kWaveMinPoints = 40;
constexpr uint32_t kWaveMaxVox = 128;

struct Rec4 {
  f32x4 r0, r1, r2, r3;
};
__device__ __forceinline__ Rec4 wave_load4(cf32x4_ptr pts4, uint32_t f, uint32_t j, uint32_t c) {
  const uint32_t last = c - 1u;
  Rec4 q;
  q.r0 = pts4[f + (j < last ? j : last)];
  q.r1 = pts4[f + (j + 1u < last ? j + 1u : last)];
  q.r2 = pts4[f + (j + 2u < last ? j + 2u : last)];
  q.r3 = pts4[f + (j + 3u < last ? j + 3u : last)];
  return q;
}

// minimum d2 of the lanes in `mine` over the records [f, f + c) -- wave-uniform f, c; the next chunk is requested before
// the current one is used (a scalar load that misses the scalar cache takes a few hundred ns)
__device__ __forceinline__ void wave_scan_bound(cf32x4_ptr pts4, uint32_t f, uint32_t c, bool mine, float qx,
                                                float qy, float qz, float& bound) {
  if (c == 0u) return;
  Rec4 cur = wave_load4(pts4, f, 0u, c);
  for (uint32_t j = 0; j < c; j += 4u) {
    const Rec4 nxt = wave_load4(pts4, f, j + 4u, c);  // (clamped: past the end it re-reads the last record)
    if (mine) {
#define MH_WB(R)                                                   \
  {                                                                \
    const float dx = R.x - qx, dy = R.y - qy, dz = R.z - qz;       \
    bound = fminf(bound, (dx * dx + dy * dy) + dz * dz);           \
  }
      MH_WB(cur.r0) MH_WB(cur.r1) MH_WB(cur.r2) MH_WB(cur.r3)
#undef MH_WB
    }
    cur = nxt;
  }
}

// first strict minimum of (d2) over the records [f, f + c) for the lanes in `need`; bidx = record index of the winner
__device__ __forceinline__ void wave_scan_best(cf32x4_ptr pts4, uint32_t f, uint32_t c, bool need, float qx,
                                               float qy, float qz, float& bd2, uint32_t& bidx) {
  if (c == 0u) return;
  const uint32_t last = c - 1u;
  Rec4 cur = wave_load4(pts4, f, 0u, c);
  for (uint32_t j = 0; j < c; j += 4u) {
    const Rec4 nxt = wave_load4(pts4, f, j + 4u, c);
    if (need) {
#define MH_WC(R, U)                                                                      \
  {                                                                                      \
    const float dx = R.x - qx, dy = R.y - qy, dz = R.z - qz;                             \
    const float d2 = (dx * dx + dy * dy) + dz * dz; /* fp32, un-fused, this order */     \
    const bool better = d2 < bd2;                                                        \
    bd2 = better ? d2 : bd2;                                                             \
    bidx = better ? f + (j + U < last ? j + U : last) : bidx;                            \
  }
      MH_WC(cur.r0, 0u) MH_WC(cur.r1, 1u) MH_WC(cur.r2, 2u) MH_WC(cur.r3, 3u)
#undef MH_WC
    }
    cur = nxt;
  }
}

// The same two scans with the records staged in the wave's LDS region as SoA x | y | z: three wave-uniform ds_read_b128
// (broadcast reads: ~100 cycles where a scalar load that misses the 16 KiB scalar cache waits ~1 us) fetch four candidates.
// Every voxel's run starts at a multiple of four records, so the reads are 16-byte aligned; the tail of the last chunk is
// padding, masked by the wave-uniform `j + U < c` tests.
constexpr uint32_t kWaveMaxRec = 640;  // padded records per wave (7.5 KiB of coordinates; C2: 90 % of the tiles fit)
struct alignas(16) WaveShared {
  float x[kWaveMaxRec], y[kWaveMaxRec], z[kWaveMaxRec];
  uint32_t lf[kWaveMaxVox], lo[kWaveMaxVox], lc[kWaveMaxVox];  // occupied voxels, compacted: first record, LDS offset, count
};

__device__ __forceinline__ void wave_scan_bound_lds(const WaveShared& sh, uint32_t o, uint32_t c, bool mine, float qx, float qy,
                                                    float qz, float& bound) {
  for (uint32_t j = 0; j < c; j += 4u) {
    const f32x4 X = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(&sh.x[o + j], 16)),
                Y = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(&sh.y[o + j], 16)),
                Z = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(&sh.z[o + j], 16));
    if (mine) {
#define MH_WB(U, CX, CY, CZ)                                        \
  if (j + U < c) {                                                  \
    const float dx = CX - qx, dy = CY - qy, dz = CZ - qz;           \
    bound = fminf(bound, (dx * dx + dy * dy) + dz * dz);            \
  }
      MH_WB(0u, X.x, Y.x, Z.x) MH_WB(1u, X.y, Y.y, Z.y) MH_WB(2u, X.z, Y.z, Z.z) MH_WB(3u, X.w, Y.w, Z.w)
#undef MH_WB
    }
  }
}
__device__ __forceinline__ void wave_scan_best_lds(const WaveShared& sh, uint32_t o, uint32_t f, uint32_t c, bool need, float qx,
                                                   float qy, float qz, float& bd2, uint32_t& bidx) {
  for (uint32_t j = 0; j < c; j += 4u) {
    const f32x4 X = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(&sh.x[o + j], 16)),
                Y = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(&sh.y[o + j], 16)),
                Z = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(&sh.z[o + j], 16));
    if (need) {
#define MH_WC(U, CX, CY, CZ)                                                             \
  if (j + U < c) { /* wave-uniform */                                                    \
    const float dx = CX - qx, dy = CY - qy, dz = CZ - qz;                                \
    const float d2 = (dx * dx + dy * dy) + dz * dz; /* fp32, un-fused, this order */     \
    const bool better = d2 < bd2;                                                        \
    bd2 = better ? d2 : bd2;                                                             \
    bidx = better ? f + j + U : bidx;                                                    \
  }
      MH_WC(0u, X.x, Y.x, Z.x) MH_WC(1u, X.y, Y.y, Z.y) MH_WC(2u, X.z, Y.z, Z.z) MH_WC(3u, X.w, Y.w, Z.w)
#undef MH_WC
    }
  }
}

// Every lane of the wave calls this with ITS query (lane `active`: it holds a point).  `sh`: this wave's LDS region; no
// workgroup barrier (LDS operations of one wave execute in order).
template <bool LDS>
__device__ __forceinline__ NNResult nn_search_wave(const MapView& m, WaveShared* shp, bool active, float qx, float qy, float qz MH_TILE_DBG_ARG) {
  NNResult r;
  r.d2 = __builtin_inff();
  r.found = false;
  r.pt = (f32x4)(0.f);
  const uint32_t lane = (uint32_t)__lane_id();
  WaveShared& sh = *shp;     // only touched under use_lds (compile-time false without LDS)
  const float lim = 1.0e6f;  // one test, no short-circuit branches: NaN and inf fail it as well
  const bool valid = active && ((int)(fabsf(qx * m.inv_vs) < lim) & (int)(fabsf(qy * m.inv_vs) < lim) & (int)(fabsf(qz * m.inv_vs) < lim));
  const int cx = valid ? voxel_of(qx, m.inv_vs, m.trunc) : 0, cy = valid ? voxel_of(qy, m.inv_vs, m.trunc) : 0,
            cz = valid ? voxel_of(qz, m.inv_vs, m.trunc) : 0;
  const int big = 0x40000000;
  const int lo_x = wave_min_i32(valid ? cx : big), lo_y = wave_min_i32(valid ? cy : big), lo_z = wave_min_i32(valid ? cz : big);
  const int hi_x = wave_max_i32(valid ? cx : -big), hi_y = wave_max_i32(valid ? cy : -big), hi_z = wave_max_i32(valid ? cz : -big);
  if (hi_x < lo_x) return r;  // no valid point in the tile (wave-uniform)
  const gslots_ptr slots4 = (gslots_ptr)m.slots;
  const gpts_ptr pts4 = (gpts_ptr)m.pts;
  const int ox = lo_x - 1, oy = lo_y - 1, oz = lo_z - 1;
  const long long ex = (long long)hi_x - lo_x + 3, ey = (long long)hi_y - lo_y + 3, ez = (long long)hi_z - lo_z + 3;
  const long long nvox_l = ex * ey * ez;
  MH_WDBG(dbg[3] = (unsigned long long)nvox_l);
  if (nvox_l > (long long)kWaveMaxVox) return valid ? nn_search_pruned(m, qx, qy, qz) : r;  // wave-uniform: lane by lane
  const uint32_t nvox = (uint32_t)nvox_l;
  const int dy = (int)ey, dz = (int)ez;
  // lane v <-> box voxels v ("a") and v + 64 ("b"); box voxel v = (ix * dy + iy) * dz + iz
  uint32_t fa = 0, ca = 0, xa = 0, fb = 0, cb = 0, xb = 0;
  {
    const float rdz = 1.0f / (float)dz, rdy = 1.0f / (float)dy;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const uint32_t v = lane + 64u * (uint32_t)half;
      if (half == 1 && nvox <= 64u) break;  // wave-uniform
      const uint32_t t = (uint32_t)(((float)v + 0.5f) * rdz);  // exact for these ranges (margin 1 / (2 dz))
      const uint32_t iz = v - t * (uint32_t)dz;
      const uint32_t ix = (uint32_t)(((float)t + 0.5f) * rdy);
      const uint32_t iy = t - ix * (uint32_t)dy;
      const unsigned long long key = pack_key(ox + (int)ix, oy + (int)iy, oz + (int)iz);
      uint32_t f = 0, c = 0;
      if (v < nvox) nn_resolve(m, slots4, key, slots4[hash_key(key) & m.mask], true, f, c);
      const uint32_t xyz = ix | (iy << 8) | (iz << 16);
      if (half == 0) { fa = f; ca = c; xa = xyz; } else { fb = f; cb = c; xb = xyz; }
    }
  }
  const cf32x4_ptr cpts = uniform_const_ptr(m.pts);
  // the box's records into LDS (when they fit): every voxel's run padded to a multiple of four records
  const uint32_t pa = (ca + 3u) & ~3u, pb = (cb + 3u) & ~3u;
  const uint32_t incl_a = wave_scan_incl(pa);
  const uint32_t tot_a = readlane_u32(incl_a, 63);
  const uint32_t incl_b = nvox > 64u ? wave_scan_incl(pb) : 0u;
  const uint32_t total = tot_a + (nvox > 64u ? readlane_u32(incl_b, 63) : 0u);
  const uint32_t oa = incl_a - pa, ob = tot_a + incl_b - pb;  // LDS offset of the lane's voxels
  const bool use_lds = LDS && total <= kWaveMaxRec;           // wave-uniform
  MH_WDBG(dbg[4] = wall_clock64(); dbg[7] = total);
  if (use_lds) {
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned long long occ_a = __ballot(ca > 0u), occ_b = __ballot(cb > 0u);
    const uint32_t n_a = (uint32_t)__popcll(occ_a), n_occ = n_a + (uint32_t)__popcll(occ_b);
    if (ca > 0u) {
      const uint32_t k = (uint32_t)__popcll(occ_a & lt);
      sh.lf[k] = fa; sh.lo[k] = oa; sh.lc[k] = ca;
    }
    if (cb > 0u) {
      const uint32_t k = n_a + (uint32_t)__popcll(occ_b & lt);
      sh.lf[k] = fb; sh.lo[k] = ob; sh.lc[k] = cb;
    }
    wave_sync_lds_nn();
    const uint32_t r16 = lane & 15u;
    for (uint32_t k = lane >> 4; k < n_occ; k += 4u) {  // sixteen lanes per occupied voxel: coalesced 16-byte records
      const uint32_t f = sh.lf[k], o = sh.lo[k], c = sh.lc[k];
      for (uint32_t j = r16; j < c; j += 16u) {
        const f32x4 rec = pts4[f + j];
        sh.x[o + j] = rec.x;
        sh.y[o + j] = rec.y;
        sh.z[o + j] = rec.z;
      }
    }
    wave_sync_lds_nn();
  }
  MH_WDBG(dbg[5] = wall_clock64());
  const int lx = cx - ox, ly = cy - oy, lz = cz - oz;                  // the lane's own voxel in box coordinates
  const uint32_t b0 = (uint32_t)((lx * dy + ly) * dz + lz);
  const Gaps gx = axis_gaps(qx, cx, m.vs, m.trunc), gy = axis_gaps(qy, cy, m.vs, m.trunc), gz = axis_gaps(qz, cz, m.vs, m.trunc);
  // pass 1: a bound from the own voxels
  float bound = __builtin_inff();
  {
    unsigned long long rem = __ballot(valid);
    while (rem) {
      const uint32_t l0 = (uint32_t)__builtin_ctzll(rem);
      const uint32_t v = readlane_u32(b0, l0);
      const bool mine = valid && b0 == v;
      rem &= ~__ballot(mine);
      const uint32_t c = v < 64u ? readlane_u32(ca, v) : readlane_u32(cb, v - 64u);
      if (use_lds) {
        const uint32_t o = v < 64u ? readlane_u32(oa, v) : readlane_u32(ob, v - 64u);
        wave_scan_bound_lds(sh, o, c, mine, qx, qy, qz, bound);
      } else {
        const uint32_t f = v < 64u ? readlane_u32(fa, v) : readlane_u32(fb, v - 64u);
        wave_scan_bound(cpts, f, c, mine, qx, qy, qz, bound);
      }
    }
  }
  MH_WDBG(dbg[6] = wall_clock64());
  // pass 2: occupied box voxels in ascending order
  float bd2 = __builtin_inff();
  uint32_t bidx = 0xFFFFFFFFu;
  int cur_vx = -1, cur_vy = -1;
  float gsx = 0.f, gsxy = 0.f;
  bool okx = false, okxy = false;
#pragma unroll
  for (int half = 0; half < 2; half++) {
    if (half == 1 && nvox <= 64u) break;
    unsigned long long occ = __ballot((half == 0 ? ca : cb) > 0u);
    while (occ) {
      const uint32_t v = (uint32_t)__builtin_ctzll(occ);
      occ &= occ - 1ull;
      const uint32_t xyz = readlane_u32(half == 0 ? xa : xb, v);
      const int vx = (int)(xyz & 0xFFu), vy = (int)((xyz >> 8) & 0xFFu), vz = (int)(xyz >> 16);
      if (vx != cur_vx) {  // wave-uniform: the x part only changes between slabs
        cur_vx = vx;
        cur_vy = -1;
        const int d = vx - lx;
        okx = (uint32_t)(d + 1) <= 2u;
        gsx = d == 0 ? 0.f : (d < 0 ? gx.s[0] : gx.s[2]);
      }
      if (vy != cur_vy) {
        cur_vy = vy;
        const int d = vy - ly;
        okxy = okx && (uint32_t)(d + 1) <= 2u;
        gsxy = gsx + (d == 0 ? 0.f : (d < 0 ? gy.s[0] : gy.s[2]));
      }
      const int d = vz - lz;
      const float lb = gsxy + (d == 0 ? 0.f : (d < 0 ? gz.s[0] : gz.s[2]));
      const bool need = valid && okxy && (uint32_t)(d + 1) <= 2u && !(lb * 0.9999f > bound);
      if (__ballot(need) == 0ull) continue;
      const uint32_t f = readlane_u32(half == 0 ? fa : fb, v), c = readlane_u32(half == 0 ? ca : cb, v);
      if (use_lds)
        wave_scan_best_lds(sh, readlane_u32(half == 0 ? oa : ob, v), f, c, need, qx, qy, qz, bd2, bidx);
      else
        wave_scan_best(cpts, f, c, need, qx, qy, qz, bd2, bidx);
      bound = fminf(bound, bd2);
    }
  }
  if (valid && bidx != 0xFFFFFFFFu) {
    r.pt = pts4[bidx];
    r.d2 = bd2;
    r.found = true;
  }
  return r;
}

