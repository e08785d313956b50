// mh_nn_flat.h -- the plan / scan matcher (round 5): the bounded correspondence search of a large layer with the
// per-point bookkeeping done ONCE per point, the per-voxel work once per candidate voxel and the per-record work once
// per record -- each at 64 work items per wave -- and the three kinds of work item handed from one phase to the next
// through the wave's slice of LDS.
//
// Why: the quad matcher (nn_search_quad, mh_nn_device.h) gives a point four lanes, and every lane of the quad repeats the
// point's bookkeeping -- fp64 transform, voxel index, slab gaps, 27 lower bounds, batch selection, key / hash / resolve /
// narrow of the probed voxel, merged-range selects: ~750 vector instructions per wave of SIXTEEN points of which ~100 test
// records (profiles/r04_match_kernel.md, VERDICT r4 "what's weak" 3).  Here a wave owns 64 consecutive scan points:
//   A1  lane = point      load l and the previous pairing, p' = (float)(R l + t), bound b0 = d2(p', old partner), voxel,
//                         gaps, the 27 lower bounds -> bit mask of candidate voxels (every voxel that can hold a record within
//                         b0: the own voxel and, for a converging alignment, 1-7 neighbours); the (point, code) pairs of the
//                         wave are written to LDS back to back (DPP prefix sum of the popcounts);
//   A2  lane = candidate  key, hash, ONE probe per lane, resolve, narrow to the hull of the quadrants of the sub-voxel
//                         index that can hold a record within b0 (quad_narrow) -> the range is cut into chunks of <= 4
//                         consecutive records, written to LDS back to back (second prefix sum);
//   B   lane = record     lane 4c+s tests record s of chunk c: d2 in the candidate arithmetic, key (d2 bits << 32 | scan
//                         position), ds_min_u64 into the point's result word (LDS atomic: the minimum does not depend on
//                         the order) -- no lane waits for a neighbour's longer list, no DPP reduction, W chunks per lane in
//                         flight;
//   C   lane = point      the winner's record from `pts`, threshold test, pairing stored (coalesced: 16 + 4 bytes per lane);
//   D   quad = point      the points the plan does not cover run nn_search_quad as before, sixteen at a time off a compacted
//                         list (ballot + prefix count): no previous pairing (ICP iteration 0: all of them), more than
//                         kFlatMaxCand candidate voxels, chunk space exhausted, or the bound not attained inside the 27-voxel
//                         block (the old partner left it: nothing beat the initial key -> once more without a bound).
// Exactness: the result of the reference's scan is the lexicographic minimum of (d2, scan position) over the 27-voxel block.
// b0 is attained by a map record, so whenever that record lies in the block the minimum has d2 <= b0; the candidate voxels
// are every voxel whose conservative lower bound does not exceed b0 and quad_narrow only drops quadrants whose every record
// is provably farther than b0 -- the set of tested records contains every record of the block with d2 <= b0, and the minimum
// over it is the block's.  Same fp32 arithmetic, same 64-bit key as nn_scan_round_quad: bit-identical pairings
// (tests/test_gpu_parity.py, tests/test_gpu_fullsize.py, tools/fuzz_bound.py with MH_MATCH=f).
#pragma once
#include "mh_nn_device.h"

namespace mh {

constexpr int kFlatLPP = 4;            // records per chunk = lanes per chunk (consecutive 16-byte records: one L1 line mostly)
#ifndef MH_FLAT_W
#define MH_FLAT_W 4
#endif
constexpr int kFlatW = MH_FLAT_W;      // chunks in flight per lane group and round: 16 x W chunks per wave and round trip
constexpr int kFlatMaxCand = 8;        // candidate voxels per point on the planned path (C2: 99.9 % of the points have <= 8)
#ifndef MH_FLAT_CHUNKS
#define MH_FLAT_CHUNKS 768
#endif
constexpr int kFlatMaxChunks = MH_FLAT_CHUNKS;  // per wave (C2: ~320)
constexpr uint32_t kFlatMaxRecords = 1u << 30;  // chunk word = first record (30 bits) | (records - 1) << 30

struct FlatWave {
  f32x4 P[64];                         // p' and the bound b0 (phase D: the bound the quad search is to start from)
  unsigned long long KB[64];           // packed key of voxel (cx-1, cy-1, cz-1)
  unsigned long long RES[64];          // best (d2 bits << 32 | scan position) so far, ds_min_u64
  uint32_t CH[kFlatMaxChunks];         // chunk: first record | (records - 1) << 30
  unsigned short CL[64 * kFlatMaxCand];  // candidate: point | code << 8
  unsigned char CHP[kFlatMaxChunks];   // the chunk's point
  unsigned char SLOWF[64];             // the plan ran out of chunk space for this point
  unsigned char SL[64];                // phase D: the points to search quad-wise, compacted
};

// count of set bits of `m` below this lane
__device__ __forceinline__ uint32_t lanes_below(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// One wave, 64 consecutive scan points starting at `i0` (lanes past `n` idle).  perm: null, or the layer is in search
// order and point i's pairing goes to perm[i].  Everything uniform (pose, thresholds, have_prev) comes in SGPRs.
__device__ __forceinline__ void match_flat_wave(FlatWave& sh, const MapView& m, const double* __restrict__ T, float thr2,
                                                float ang2, bool have_prev, const float* __restrict__ lx,
                                                const float* __restrict__ ly, const float* __restrict__ lz, uint32_t n,
                                                uint32_t i0, float4* __restrict__ pair_q, uint32_t* __restrict__ pair_gidx,
                                                const uint32_t* __restrict__ perm) {
  const uint32_t lane = (uint32_t)__lane_id();
  const uint32_t i = i0 + lane;
  const bool in = i < n;
  const uint32_t ic = in ? i : n - 1;
  const uint32_t o = perm ? G(perm)[ic] : ic;
  const gslots_ptr slots4 = (gslots_ptr)m.slots;
  const gpts_ptr pts4 = (gpts_ptr)m.pts;
  const gpts_ptr spts = (gpts_ptr)m.pts_q;
  // ---- A1: the point ------------------------------------------------------------------------------------------------
  const float x = G(lx)[ic], y = G(ly)[ic], z = G(lz)[ic];
  f32x4 prev = (f32x4){0.f, 0.f, 0.f, __builtin_inff()};
  if (have_prev) prev = G(reinterpret_cast<const f32x4*>(pair_q))[o];  // grid-uniform branch
  float px, py, pz;
  transform_point(T, x, y, z, px, py, pz);
  float b0 = __builtin_inff();
  if (prev.w < __builtin_inff()) {
    const float dx = prev.x - px, dy = prev.y - py, dz = prev.z - pz;
    b0 = (dx * dx + dy * dy) + dz * dz;  // the candidate arithmetic
  }
  const float lim = 1.0e6f;
  const bool okrange = ((int)(fabsf(px * m.inv_vs) < lim) & (int)(fabsf(py * m.inv_vs) < lim) & (int)(fabsf(pz * m.inv_vs) < lim)) != 0;
  bool planned = in && okrange && b0 < __builtin_inff();
  uint32_t cmask = 0;
  unsigned long long kbase = 0;
  if (__ballot(planned) != 0ull) {  // wave-uniform (iteration 0: nobody)
    const int cx = voxel_of(px, m.inv_vs, m.trunc), cy = voxel_of(py, m.inv_vs, m.trunc), cz = voxel_of(pz, m.inv_vs, m.trunc);
    kbase = pack_key(cx - 1, cy - 1, cz - 1);
    const Gaps gx = axis_gaps(px, cx, m.vs, m.trunc), gy = axis_gaps(py, cy, m.vs, m.trunc), gz = axis_gaps(pz, cz, m.vs, m.trunc);
    cmask = 1u << 13;
#pragma unroll
    for (int c = 0; c < 27; c++) {
      if (c == 13) continue;
      const int ix = c / 9, iy = (c / 3) % 3, iz = c % 3;
      const float sx = ix == 1 ? 0.f : gx.s[ix], sy = iy == 1 ? 0.f : gy.s[iy], sz = iz == 1 ? 0.f : gz.s[iz];
      const float lb = ((sx + sy) + sz) * 0.9999f;  // quad_bounds' expression
      cmask |= (!(lb > b0)) ? (1u << c) : 0u;
    }
    if (!planned) cmask = 0;
    if (__builtin_popcount(cmask) > kFlatMaxCand) {  // a loose bound near a voxel corner: the quad search, with the bound
      planned = false;
      cmask = 0;
    }
  }
  const uint32_t ncand = (uint32_t)__builtin_popcount(cmask);
  const uint32_t cincl = wave_scan_incl(ncand);
  const uint32_t n_cands = (uint32_t)__builtin_amdgcn_readlane((int)cincl, 63);
  sh.P[lane] = (f32x4){px, py, pz, b0};
  uint32_t nvalid = 0;  // chunks written (wave-uniform)
  if (n_cands) {        // wave-uniform
    sh.KB[lane] = kbase;
    sh.RES[lane] = ((unsigned long long)__float_as_uint(b0) << 32) | 0xFFFFFFFFull;
    sh.SLOWF[lane] = 0;
    {
      uint32_t mm = cmask, at = cincl - ncand;
#pragma unroll
      for (int j = 0; j < kFlatMaxCand; j++) {
        if (mm) {
          const uint32_t code = (uint32_t)__builtin_ctz(mm);
          mm &= mm - 1;
          sh.CL[at + (uint32_t)j] = (unsigned short)(lane | (code << 8));
        }
      }
    }
    wave_sync_lds_nn();
    // ---- A2: the candidate voxel ---------------------------------------------------------------------------------------
    uint32_t nch_total = 0;
    bool overflowed = false;
    for (uint32_t base = 0; base < n_cands; base += 64u) {
      const uint32_t c = base + lane;
      const bool act = c < n_cands;
      const uint32_t e = sh.CL[act ? c : 0u];
      const uint32_t p = e & 63u;
      const int code = (int)(e >> 8);
      const f32x4 P = sh.P[p];
      const unsigned long long key = nn_key_of(sh.KB[p], code);
      const u32x4 sl = slots4[hash_key(key) & m.mask];
      uint32_t f, cnt, qv;
      nn_resolve(m, slots4, key, sl, act, f, cnt, &qv);
      quad_narrow(m, qv, code, P.x, P.y, P.w, f, cnt);
      const uint32_t nchunks = (cnt + (uint32_t)kFlatLPP - 1u) / (uint32_t)kFlatLPP;
      const uint32_t incl = wave_scan_incl(nchunks);
      const uint32_t pos = nch_total + incl - nchunks;
      const bool unfit = nchunks != 0u && (overflowed || pos + nchunks > (uint32_t)kFlatMaxChunks);
      const unsigned long long ub = __ballot(unfit);
      if (ub != 0ull && !overflowed) {  // wave-uniform: everything from the first lane that does not fit goes to phase D
        nvalid = (uint32_t)__builtin_amdgcn_readlane((int)pos, __builtin_ctzll(ub));
        overflowed = true;
      }
      if (unfit) sh.SLOWF[p] = 1;
      if (!unfit) {
        for (uint32_t k = 0; k < nchunks; k++) {
          const uint32_t rem = cnt - (uint32_t)kFlatLPP * k;
          sh.CH[pos + k] = (f + (uint32_t)kFlatLPP * k) | (((rem < (uint32_t)kFlatLPP ? rem : (uint32_t)kFlatLPP) - 1u) << 30);
          sh.CHP[pos + k] = (unsigned char)p;
        }
      }
      nch_total += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (!overflowed) nvalid = nch_total;
    wave_sync_lds_nn();
    // ---- B: the record ---------------------------------------------------------------------------------------------------
    const uint32_t grp = lane >> 2, sub = lane & 3u;
    for (uint32_t t0 = 0; t0 < nvalid; t0 += 16u * (uint32_t)kFlatW) {
      f32x4 rec[kFlatW], Pq[kFlatW];
      uint32_t pp[kFlatW];
      bool valid[kFlatW];
#pragma unroll
      for (int u = 0; u < kFlatW; u++) {
        const uint32_t t = t0 + 16u * (uint32_t)u + grp;
        const bool ok = t < nvalid;
        const uint32_t tt = ok ? t : 0u;  // (chunk 0 exists: nvalid > 0)
        const uint32_t ch = sh.CH[tt];
        pp[u] = sh.CHP[tt];
        const uint32_t last = ch >> 30;
        valid[u] = ok && sub <= last;
        rec[u] = spts[(ch & 0x3FFFFFFFu) + (sub < last ? sub : last)];  // clamped into the chunk: no load behind a branch
        Pq[u] = sh.P[pp[u]];
      }
#pragma unroll
      for (int u = 0; u < kFlatW; u++) {
        const float dx = rec[u].x - Pq[u].x, dy = rec[u].y - Pq[u].y, dz = rec[u].z - Pq[u].z;
        const float d2 = (dx * dx + dy * dy) + dz * dz;  // fp32, un-fused, this order (bit-exact with the oracle)
        const unsigned long long k = ((unsigned long long)__float_as_uint(d2) << 32) | __float_as_uint(rec[u].w);
        const unsigned long long kb = ((unsigned long long)__float_as_uint(Pq[u].w) << 32) | 0xFFFFFFFFull;
        if (valid[u] && k < kb)
          (void)__hip_atomic_fetch_min(&sh.RES[pp[u]], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    wave_sync_lds_nn();
  }
  // ---- C: the pairing ----------------------------------------------------------------------------------------------------
  bool slow = in && !planned;
  float b0s = b0;  // the bound phase D starts from
  if (n_cands) {
    const unsigned long long res = sh.RES[lane];
    const bool spilled = sh.SLOWF[lane] != 0;
    const uint32_t idx = (uint32_t)res;
    if (planned && !spilled && idx != 0xFFFFFFFFu) {
      const f32x4 w = pts4[idx];
      const float d2 = __uint_as_float((uint32_t)(res >> 32));
      const float n2 = (px * px + py * py) + pz * pz;
      const bool ok = d2 < thr2 + ang2 * n2;
      G(reinterpret_cast<f32x4*>(pair_q))[o] = (f32x4){w.x, w.y, w.z, d2};
      G(pair_gidx)[o] = ok ? __float_as_uint(w.w) : kNoMatch;
    } else if (planned) {
      slow = true;
      if (!spilled) b0s = __builtin_inff();  // the bound was not attained inside the block: once more, without it
    }
  }
  // ---- D: what the plan does not cover, quad-wise ---------------------------------------------------------------------
  const unsigned long long sm = __ballot(slow);
  if (sm == 0ull) return;  // wave-uniform
  const uint32_t ns = (uint32_t)__builtin_popcountll(sm);
  wave_sync_lds_nn();  // (phase B's readers of P are done)
  if (slow) {
    sh.SL[lanes_below(sm)] = (unsigned char)lane;
    sh.P[lane] = (f32x4){px, py, pz, b0s};
  }
  wave_sync_lds_nn();
  const uint32_t grp = lane >> 2, sub = lane & 3u;
  for (uint32_t g = 0; g < ns; g += 16u) {
    const uint32_t k = g + grp;
    if (k < ns) {  // whole quads
      const uint32_t p = sh.SL[k];
      const f32x4 P = sh.P[p];
      const NNResult r = nn_search_quad(m, sub, P.x, P.y, P.z, P.w);
      if (sub == 0u) {
        const uint32_t ip = i0 + p;
        const uint32_t op = perm ? G(perm)[ip] : ip;
        const float n2 = (P.x * P.x + P.y * P.y) + P.z * P.z;
        const bool ok = r.found && (r.d2 < thr2 + ang2 * n2);
        G(reinterpret_cast<f32x4*>(pair_q))[op] = (f32x4){r.pt.x, r.pt.y, r.pt.z, r.d2};
        G(pair_gidx)[op] = ok ? __float_as_uint(r.pt.w) : kNoMatch;
      }
    }
  }
}

}  // namespace mh
