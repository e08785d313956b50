// mh_nn_device.h -- device-side building blocks of the correspondence search and of the Gauss-Newton
// accumulation.  Everything that decides WHICH point pairs (transform, voxel index, fp32 distance,
// comparison order) is written un-fused and in a fixed order so that the result is bit-identical to
// the CPU restatement of the reference algorithm (this file is compiled with -ffp-contract=off).
#pragma once
#include "mh_internal.h"
#ifdef MH_CARRY_WINNER
// (round-4 experiment, measured slower: profiles/r04_match_kernel.md.  Since the scans read the sub-voxel index's copy of the
//  records -- pts_q, whose w is the record's POSITION, not its source index -- the carried record would put positions into
//  pair_gidx.  Kept in the history, not buildable)
#error "MH_CARRY_WINNER is not supported any more: nn_scan_round_quad reads pts_q records (w = scan position)"
#endif

namespace mh {

constexpr uint32_t kNoMatch = 0xFFFFFFFFu;

// (An XCD-aware block order -- workgroup b runs on XCD b % 8; hand each XCD a contiguous range of the scan so that its
// 4 MiB L2 only sees 1/8 of the map -- was measured twice, for the one-lane and for the quad kernel: 0 % and -10 %.
// The match kernels do not wait for L2 misses but for L1-miss round trips, and a contiguous range puts the crowded
// part of the scene on one XCD.  The remap was removed.)

// native clang vectors: a plain dwordx4 load into registers (HIP's uint4/float4 are union structs whose
// copies become memcpy's that keep arrays of them in scratch memory)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// Global-address-space pointers, spelled out: a pointer that reaches a kernel through a descriptor in memory (BatchJob, the
// lock-step launches) is GENERIC to the compiler and every access through it becomes a flat_load (both memory counters,
// the aperture check, no scalar path); only pointers that are kernel arguments themselves get the global space inferred.
// G(p) casts (a no-op for the latter); the table and record pointers of a MapView have types of their own.
#define MH_AS_GLOBAL __attribute__((address_space(1)))
typedef const u32x4 MH_AS_GLOBAL* gslots_ptr;
typedef const f32x4 MH_AS_GLOBAL* gpts_ptr;
template <class T>
__device__ __forceinline__ T MH_AS_GLOBAL* G(T* p) { return (T MH_AS_GLOBAL*)p; }

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

// The count word of a hash slot.  Voxels that carry a sub-voxel index (k_build_qidx: at most 31 records) keep the quadrants'
// boundaries in the same word -- bit 31 | b3 << 18 | b2 << 13 | b1 << 8 | count -- so that the probe that finds a voxel also
// brings what narrows its scan; everybody else reads the count through this.
__device__ __forceinline__ uint32_t slot_count(uint32_t w) { return (w & 0x80000000u) ? (w & 0xFFu) : w; }

struct NNResult {
  f32x4 pt;   // nearest map point {x,y,z,src}
  float d2;
  bool found;
#ifdef MH_CARRY_WINNER
  bool writer;  // nn_search_quad: this lane of the quad holds the winner's record (r.pt is valid in this lane only)
#endif
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
  const float lim = 1.0e6f;  // one test, no short-circuit branches: NaN and inf fail it as well
  if (!((int)(fabsf(qx * m.inv_vs) < lim) & (int)(fabsf(qy * m.inv_vs) < lim) & (int)(fabsf(qz * m.inv_vs) < lim))) return r;
  const int cx = voxel_of(qx, m.inv_vs, m.trunc), cy = voxel_of(qy, m.inv_vs, m.trunc), cz = voxel_of(qz, m.inv_vs, m.trunc);
  const unsigned long long kbase = pack_key(cx - 1, cy - 1, cz - 1);
  const gslots_ptr slots4 = (gslots_ptr)m.slots;  // one dwordx4 per slot
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
        cnt += slot_count(sl.w);
      }
    }
    first9[col] = first;
    cnt9[col] = cnt;
  }
#pragma unroll
  for (int col = 0; col < 9; col++) {
    const uint32_t first = first9[col], cnt = cnt9[col];
    const gpts_ptr p = (gpts_ptr)m.pts + first;
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

// -------------------------------------------------------------------------------------------------
// Exact branch-and-bound version of nn_single_search (the production path).
//
// The reference scans all 27 voxels; the RESULT is the lexicographic minimum of (d2, scan position)
// over the candidates, and because point records are stored in scan order (ascending packed key =
// x outer / y middle / z inner, then insertion order) "scan position" is simply the record index.
// That minimum does not change if a voxel is skipped whose every point is provably farther than
// the best candidate found so far.  So: scan the query's own voxel first, then visit a neighbour
// voxel only if a conservative lower bound of its distance can still beat (or tie) the current best:
//     lb = gap_x^2 + gap_y^2 + gap_z^2 ,  gap = distance from q to the voxel's coordinate slab,
// with the slab widened by a few ulps of the coordinate magnitude and the test relaxed by 1e-4, so
// that fp32 rounding of either side can never prune a candidate with d2 <= best.  On the C2 workload
// this evaluates ~35 candidates per point instead of ~175 and probes ~9 hash slots instead of 27,
// with bit-identical output (tests/test_gpu_parity.py, incl. lattice/tie and boundary cases).
// -------------------------------------------------------------------------------------------------
struct Gaps {
  float s[3];  // squared conservative gap to the slabs of voxel c-1, c, c+1 along one axis
};

__device__ __forceinline__ Gaps axis_gaps(float q, int c, float vs, uint32_t trunc) {
  // coordinate range of voxel v: floor mode [v*vs, (v+1)*vs); trunc mode: v>0 same, v<0 ((v-1)*vs, v*vs],
  // v==0 (-vs, vs).  hi(c-1) and lo(c+1) are what we need.
  const int vm = c - 1, vp = c + 1;
  const float hi_m = (float)((!trunc || vm >= 0) ? vm + 1 : vm) * vs;
  const float lo_p = (float)((!trunc || vp > 0) ? vp : vp - 1) * vs;
  const float margin = 1.0e-6f * ((float)(c < 0 ? -c : c) + 2.0f) * vs;  // >= 8 ulp of the coordinate
  const float gm = fmaxf(0.f, (q - hi_m) - margin);
  const float gp = fmaxf(0.f, (lo_p - q) - margin);
  Gaps g;
  g.s[0] = gm * gm;
  g.s[1] = 0.f;
  g.s[2] = gp * gp;
  return g;
}

struct NNBest {
  float d2;
  uint32_t idx;  // record index in MapView::pts (== scan position)
};

__device__ __forceinline__ void nn_consider(const f32x4& c, uint32_t idx, float qx, float qy, float qz, NNBest& b) {
  const float dx = c.x - qx, dy = c.y - qy, dz = c.z - qz;
  const float d2 = (dx * dx + dy * dy) + dz * dz;  // fp32, un-fused, this order (bit-exact with the oracle)
  if (d2 < b.d2 || (d2 == b.d2 && idx < b.idx)) {
    b.d2 = d2;
    b.idx = idx;
  }
}

// all records of one voxel, four loads in flight (offsets clamped into the run instead of predicated: a load inside
// an `if` makes hipcc wait for it at the end of the branch, which would serialise the round trips; eight in flight
// measured no faster and costs a wave of occupancy per SIMD)
__device__ __forceinline__ void nn_scan_voxel(gpts_ptr pts, uint32_t first, uint32_t cnt, float qx,
                                              float qy, float qz, NNBest& b) {
  for (uint32_t j = 0; j < cnt; j += 4) {
    const gpts_ptr p = pts + (first + j);
    const uint32_t rem = cnt - j;  // >= 1
    f32x4 c[4];
#pragma unroll
    for (int u = 0; u < 4; u++) c[u] = p[(uint32_t)u < rem ? u : 0];
#pragma unroll
    for (int u = 0; u < 4; u++)
      if ((uint32_t)u < rem) nn_consider(c[u], first + j + u, qx, qy, qz, b);
  }
}

// conservative lower bound of the squared distance from q to neighbour voxel `code` (= ix*9+iy*3+iz)
__device__ __forceinline__ float nn_lower_bound(int code, const Gaps& gx, const Gaps& gy, const Gaps& gz) {
  const int ix = (code * 57) >> 9;  // code / 9 for code < 27
  const int r = code - 9 * ix;
  const int iy = (r * 11) >> 5;     // r / 3 for r < 9
  const int iz = r - 3 * iy;
  // (one-level selects and a max of two non-negative values: hipcc turns the nested form into exec-mask branches)
  const float sx = fmaxf(ix == 0 ? gx.s[0] : 0.f, ix == 2 ? gx.s[2] : 0.f);
  const float sy = fmaxf(iy == 0 ? gy.s[0] : 0.f, iy == 2 ? gy.s[2] : 0.f);
  const float sz = fmaxf(iz == 0 ? gz.s[0] : 0.f, iz == 2 ? gz.s[2] : 0.f);
  return (sx + sy) + sz;
}

__device__ __forceinline__ unsigned long long nn_key_of(unsigned long long kbase, int code) {
  const int ix = (code * 57) >> 9;
  const int r = code - 9 * ix;
  const int iy = (r * 11) >> 5;
  const int iz = r - 3 * iy;
  return kbase + ((unsigned long long)ix << 42) + ((unsigned long long)iy << 21) + (unsigned long long)iz;
}

__device__ __forceinline__ void nn_visit(const MapView& m, gslots_ptr slots4, gpts_ptr pts4,
                                         unsigned long long key, u32x4 sl, float qx, float qy, float qz, NNBest& b) {
  unsigned long long sk = ((unsigned long long)sl.y << 32) | sl.x;
  if (sk != key && sk != kEmptyKey) {  // rare: linear probing past a collision
    uint32_t h = hash_key(key) & m.mask;
    do {
      h = (h + 1) & m.mask;
      sl = slots4[h];
      sk = ((unsigned long long)sl.y << 32) | sl.x;
    } while (sk != key && sk != kEmptyKey);
  }
  if (sk == key) nn_scan_voxel(pts4, sl.z, slot_count(sl.w), qx, qy, qz, b);
}

__device__ __forceinline__ NNResult nn_search_pruned(const MapView& m, float qx, float qy, float qz) {
  NNResult r;
  r.d2 = __builtin_inff();
  r.found = false;
  r.pt = (f32x4)(0.f);
  const float lim = 1.0e6f;  // one test, no short-circuit branches: NaN and inf fail it as well
  if (!((int)(fabsf(qx * m.inv_vs) < lim) & (int)(fabsf(qy * m.inv_vs) < lim) & (int)(fabsf(qz * m.inv_vs) < lim))) return r;
  const int cx = voxel_of(qx, m.inv_vs, m.trunc), cy = voxel_of(qy, m.inv_vs, m.trunc), cz = voxel_of(qz, m.inv_vs, m.trunc);
  const unsigned long long kbase = pack_key(cx - 1, cy - 1, cz - 1);
  const gslots_ptr slots4 = (gslots_ptr)m.slots;
  const gpts_ptr pts4 = (gpts_ptr)m.pts;
  const float vs = m.vs;  // only used for bounds, which carry their own safety margin
  const Gaps gx = axis_gaps(qx, cx, vs, m.trunc), gy = axis_gaps(qy, cy, vs, m.trunc), gz = axis_gaps(qz, cz, vs, m.trunc);
  NNBest b;
  b.d2 = __builtin_inff();
  b.idx = 0xFFFFFFFFu;
  // 1. the query's own voxel (code 13)
  {
    const unsigned long long key = nn_key_of(kbase, 13);
    nn_visit(m, slots4, pts4, key, slots4[hash_key(key) & m.mask], qx, qy, qz, b);
  }
  // 2. the neighbours that can still hold a candidate with d2 <= best: bit `code` of `mask`
  uint32_t mask = 0;

#pragma unroll
  for (int c = 0; c < 27; c++) {
    if (c == 13) continue;
    const float lb = (gx.s[c / 9] + gy.s[(c / 3) % 3]) + gz.s[c % 3];
    if (!(lb * 0.9999f > b.d2)) mask |= 1u << c;
  }
  // faces first, then edges, then corners: nearer voxels tighten the bound for the farther ones
  const uint32_t kFaces = (1u << 4) | (1u << 10) | (1u << 12) | (1u << 14) | (1u << 16) | (1u << 22);
  const uint32_t kCorners = (1u << 0) | (1u << 2) | (1u << 6) | (1u << 8) | (1u << 18) | (1u << 20) | (1u << 24) | (1u << 26);
  const uint32_t kEdges = 0x07FFFFFFu & ~(kFaces | kCorners | (1u << 13));
#pragma unroll 1
  for (int cls = 0; cls < 3; cls++) {
    uint32_t mm = mask & (cls == 0 ? kFaces : (cls == 1 ? kEdges : kCorners));
    while (mm) {
      // up to four probes in flight
      int c0 = __builtin_ctz(mm), c1 = -1, c2 = -1, c3 = -1;
      mm &= mm - 1;
      if (mm) { c1 = __builtin_ctz(mm); mm &= mm - 1; }
      if (mm) { c2 = __builtin_ctz(mm); mm &= mm - 1; }
      if (mm) { c3 = __builtin_ctz(mm); mm &= mm - 1; }
      const unsigned long long k0 = nn_key_of(kbase, c0), k1 = nn_key_of(kbase, c1 < 0 ? 0 : c1),
                               k2 = nn_key_of(kbase, c2 < 0 ? 0 : c2), k3 = nn_key_of(kbase, c3 < 0 ? 0 : c3);
      const u32x4 s0 = slots4[hash_key(k0) & m.mask], s1 = slots4[hash_key(k1) & m.mask],
                  s2 = slots4[hash_key(k2) & m.mask], s3 = slots4[hash_key(k3) & m.mask];  // unconditional: 4 in flight
      if (!(nn_lower_bound(c0, gx, gy, gz) * 0.9999f > b.d2)) nn_visit(m, slots4, pts4, k0, s0, qx, qy, qz, b);
      if (c1 >= 0 && !(nn_lower_bound(c1, gx, gy, gz) * 0.9999f > b.d2)) nn_visit(m, slots4, pts4, k1, s1, qx, qy, qz, b);
      if (c2 >= 0 && !(nn_lower_bound(c2, gx, gy, gz) * 0.9999f > b.d2)) nn_visit(m, slots4, pts4, k2, s2, qx, qy, qz, b);
      if (c3 >= 0 && !(nn_lower_bound(c3, gx, gy, gz) * 0.9999f > b.d2)) nn_visit(m, slots4, pts4, k3, s3, qx, qy, qz, b);
    }
  }
  if (b.idx != 0xFFFFFFFFu) {
    r.pt = pts4[b.idx];
    r.d2 = b.d2;
    r.found = true;
  }
  return r;
}


// -------------------------------------------------------------------------------------------------
// NearestNeighborsCapable::nn_multiple_search(q, k) [U] ("same scan keeping k best sorted"; used by
// Matcher_Points_DistanceThreshold with pairingsPerPoint > 1, rgbd.yaml:135-141): the k smallest (d2, scan position) of the
// 27-voxel block.  One lane per point, the eight smallest keys (d2 bits << 32 | record index: unsigned order = the reference's
// order, ties to the earlier scan position) kept sorted in registers by a chain of compare-exchanges, voxels pruned against
// the k-th smallest so far with the bound of nn_search_pruned.  Not a hot path of either target pipeline: exactness first.
// -------------------------------------------------------------------------------------------------
constexpr int kMaxKnn = 8;         // Matcher_Points_DistanceThreshold::pairingsPerPoint (MH_MAX_PAIRINGS_PER_POINT)
constexpr int kMaxPlaneKnn = 16;   // Matcher_Point2Plane::knn on a plain point map (MH_MAX_PLANE_KNN; rgbd.yaml:148 uses 10)
typedef unsigned long long knnkey_t;
template <int CAP>
__device__ __forceinline__ void knn_insert(knnkey_t (&best)[CAP], knnkey_t key) {
  if (!(key < best[CAP - 1])) return;
  best[CAP - 1] = key;
#pragma unroll
  for (int t = CAP - 1; t > 0; t--) {
    const knnkey_t a = best[t - 1], b = best[t];
    const bool sw = b < a;
    best[t - 1] = sw ? b : a;
    best[t] = sw ? a : b;
  }
}
template <int CAP>
__device__ __forceinline__ knnkey_t knn_select(const knnkey_t (&best)[CAP], uint32_t r) {
  knnkey_t v = best[0];
#pragma unroll
  for (int t = 1; t < CAP; t++) v = r == (uint32_t)t ? best[t] : v;
  return v;
}
template <int CAP>
__device__ __forceinline__ void knn_visit(const MapView& m, gslots_ptr slots4, gpts_ptr pts4, unsigned long long key, float qx,
                                          float qy, float qz, knnkey_t (&best)[CAP]) {
  uint32_t h = hash_key(key) & m.mask;
  u32x4 sl = slots4[h];
  unsigned long long sk = ((unsigned long long)sl.y << 32) | sl.x;
  while (sk != key && sk != kEmptyKey) {  // linear probing past a collision
    h = (h + 1) & m.mask;
    sl = slots4[h];
    sk = ((unsigned long long)sl.y << 32) | sl.x;
  }
  if (sk != key) return;
  const uint32_t n_rec = slot_count(sl.w);
  for (uint32_t j = 0; j < n_rec; j++) {
    const f32x4 c = pts4[sl.z + j];
    const float dx = c.x - qx, dy = c.y - qy, dz = c.z - qz;
    const float d2 = (dx * dx + dy * dy) + dz * dz;  // fp32, un-fused, this order (bit-exact with the oracle)
    knn_insert<CAP>(best, ((knnkey_t)__float_as_uint(d2) << 32) | (knnkey_t)(sl.z + j));
  }
}
// best[] ascending on return; entries never filled stay ~0
template <int CAP>
__device__ __forceinline__ void nn_search_kbest(const MapView& m, float qx, float qy, float qz, uint32_t k,
                                                knnkey_t (&best)[CAP]) {
#pragma unroll
  for (int t = 0; t < CAP; t++) best[t] = ~0ull;
  const float lim = 1.0e6f;
  if (!((int)(fabsf(qx * m.inv_vs) < lim) & (int)(fabsf(qy * m.inv_vs) < lim) & (int)(fabsf(qz * m.inv_vs) < lim))) return;
  const int cx = voxel_of(qx, m.inv_vs, m.trunc), cy = voxel_of(qy, m.inv_vs, m.trunc), cz = voxel_of(qz, m.inv_vs, m.trunc);
  const unsigned long long kbase = pack_key(cx - 1, cy - 1, cz - 1);
  const gslots_ptr slots4 = (gslots_ptr)m.slots;
  const gpts_ptr pts4 = (gpts_ptr)m.pts;
  const Gaps gx = axis_gaps(qx, cx, m.vs, m.trunc), gy = axis_gaps(qy, cy, m.vs, m.trunc), gz = axis_gaps(qz, cz, m.vs, m.trunc);
  knn_visit<CAP>(m, slots4, pts4, nn_key_of(kbase, 13), qx, qy, qz, best);
#pragma unroll 1
  for (int c = 0; c < 27; c++) {
    if (c == 13) continue;
    // the k-th smallest distance so far (+inf while fewer than k were found: the key's high word is then 0xFFFFFFFF = NaN,
    // and `lb > NaN` is false -- the voxel is visited)
    const float bound = __uint_as_float((uint32_t)(knn_select<CAP>(best, k - 1) >> 32));
    if (nn_lower_bound(c, gx, gy, gz) * 0.9999f > bound) continue;
    knn_visit<CAP>(m, slots4, pts4, nn_key_of(kbase, c), qx, qy, qz, best);
  }
}

// {first, count} of voxel `key` given the slot its hash points at; count 0 when absent or not wanted
// (raw: the slot's count word as stored -- with the quadrants' boundaries of an indexed voxel, slot_count() -- wherever linear
// probing found the voxel)
__device__ __forceinline__ void nn_resolve(const MapView& m, gslots_ptr slots4, unsigned long long key, u32x4 sl,
                                           bool want, uint32_t& first, uint32_t& cnt, uint32_t* raw = nullptr) {
  first = 0;
  cnt = 0;
  if (raw) *raw = 0u;
  if (!want) return;
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
    first = sl.z;
    cnt = slot_count(sl.w);
    if (raw) *raw = sl.w;
  }
}

// -------------------------------------------------------------------------------------------------
// Four lanes per scan point (a DPP quad): the same exact branch-and-bound search, with the point's work spread over
// its quad.  Lane `sub` of the quad issues the probe of the batch's voxel #sub (one load instead of four) and reads
// every fourth candidate of the merged record ranges; the quad agrees on the best (d2, record) with two quad_perm
// steps after every scan, so its four lanes always take the same branches.  Per lane that is a quarter of the load
// instructions and a quarter of the scan round trips; a quad's loads hit four consecutive 16-byte records, which the
// vector L1 serves ~3.6x faster than four scattered ones (tools/gather_bench.hip: 888 vs 246 M loads/ms).
//
// What the measurements behind this design say (C2, MI355X; per-wave wall_clock64 traces, PMC passes, A/B runs):
//   * a launch lasts as long as its slowest wave; a wave costs ~7.5 us + ~0.45 us per dependent round trip of its
//     slowest lane (every round trip of a 64-lane wave contains an L1 miss, so it always pays L2/Infinity-Cache latency);
//   * with one lane per point (nn_search_pruned) the slowest lanes chain up to ~90 round trips (57 us per launch);
//     merging the scans of a probe batch and splitting a point over a quad brings that to ~26 (28 us incl. k_accum);
//   * what did NOT help: 8-wide scans or twice as many loads in flight for crowded batches (registers -> occupancy),
//     probing all 26 neighbours up front (more loads and VALU than the saved round trips are worth on C2), forcing
//     the loads of a round to be issued unconditionally, an XCD-contiguous block order, other workgroup sizes.
// -------------------------------------------------------------------------------------------------
#ifndef MH_QUAD_W
#define MH_QUAD_W 4  // candidates per lane and round trip: 4 x 4 = 16.  Round 2 (search bounded by the previous pairing, 58
                     // VGPRs at W = 3): W = 4 still fits 64 VGPRs / 8 waves per SIMD without scratch and measured 8.16 us
                     // per scan in lock step against 8.45 (W = 3), 8.60 (W = 5 at 70 VGPRs / 7 waves), 9.4 (W = 5 with
                     // scratch, W = 6): see MH_QUAD_WAVES in mh_icp.hip
#endif
constexpr int kQuadW = MH_QUAD_W;

template <int CTRL>
__device__ __forceinline__ uint32_t quad_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
template <int L>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) { return quad_u32<L * 0x55>(v); }  // quad_perm:[L,L,L,L]

// (d2 bits << 32 | record index): for the non-negative d2 here the unsigned order of this key IS the lexicographic
// (d2, scan position) order the reference's first-strict-minimum rule induces; one 64-bit compare per candidate.
typedef unsigned long long nnkey_t;
constexpr nnkey_t kNNKeyNone = 0x7F800000FFFFFFFFull;  // (+inf, no record)
__device__ __forceinline__ float nnkey_d2(nnkey_t k) { return __uint_as_float((uint32_t)(k >> 32)); }
__device__ __forceinline__ uint32_t nnkey_idx(nnkey_t k) { return (uint32_t)k; }

__device__ __forceinline__ nnkey_t quad_min_key(nnkey_t k) {
#define MH_QUAD_STEP(CTRL)                                                                   \
  {                                                                                          \
    const nnkey_t o = ((nnkey_t)quad_u32<CTRL>((uint32_t)(k >> 32)) << 32) | quad_u32<CTRL>((uint32_t)k); \
    k = o < k ? o : k;                                                                       \
  }
  MH_QUAD_STEP(0xB1)  // quad_perm:[1,0,3,2]
  MH_QUAD_STEP(0x4E)  // quad_perm:[2,3,0,1]
#undef MH_QUAD_STEP
  return k;
}

// One round trip: W records per lane of the merged ranges.
#ifdef MH_CARRY_WINNER
// (experiment, VERDICT r3 item 2c "the winner's xyz carried with the key"): every lane keeps the RECORD of its own best
// candidate next to the key; after the quad has agreed on the best key, the one lane whose own best it is holds the record
// in registers and writes the pairing -- the dependent fetch of the winner's record at the end of the search disappears.
#define MH_CARRY_ARG , f32x4& brec, nnkey_t& mine
#define MH_CARRY_PASS , brec, mine
#else
#define MH_CARRY_ARG
#define MH_CARRY_PASS
#endif
template <int NV, int W>
__device__ __forceinline__ nnkey_t nn_scan_round_quad(gpts_ptr pts4, const uint32_t (&start)[NV],
                                                      const uint32_t (&pre)[NV + 1], uint32_t t0, uint32_t sub, float qx,
                                                      float qy, float qz, nnkey_t best MH_CARRY_ARG) {
  const uint32_t total = pre[NV];
  f32x4 c[W];
  uint32_t ri[W];
  bool valid[W];
#pragma unroll
  for (int u = 0; u < W; u++) {
    const uint32_t tu = t0 + 4u * (uint32_t)u + sub;
    valid[u] = tu < total;
    const uint32_t t = valid[u] ? tu : total - 1;  // clamped into the ranges: no load sits behind a data-dependent branch in the source
    uint32_t off = start[0];                        // start[v] = first[v] - pre[v]: record = t + start[voxel of t]
#pragma unroll
    for (int v = 1; v < NV; v++) off = t >= pre[v] ? start[v] : off;
    ri[u] = t + off;
    c[u] = pts4[ri[u]];
  }
#pragma unroll
  for (int u = 0; u < W; u++) {
    const float dx = c[u].x - qx, dy = c[u].y - qy, dz = c[u].z - qz;
    const float d2 = (dx * dx + dy * dy) + dz * dz;  // fp32, un-fused, this order (bit-exact with the oracle)
    // (measured: hipcc turns this select into a branch and sinks the load of a lane past the end under it; forcing
    // straight-line code with every load issued -- masks, sched_barrier, inline-asm loads -- was 30-40 % SLOWER on C2:
    // the clamped duplicate loads cost more in the texture-address path than the extra waits do)
    // (the records come in the sub-voxel index's order, MapView::pts_q: the scan position the tie-break needs rides in w)
    const nnkey_t k = valid[u] ? (((nnkey_t)__float_as_uint(d2) << 32) | __float_as_uint(c[u].w)) : kNNKeyNone;
#ifdef MH_CARRY_WINNER
    const bool better = k < mine;
    mine = better ? k : mine;
    brec = better ? c[u] : brec;
#endif
    best = k < best ? k : best;
  }
  return best;
}

template <int NV>
__device__ __forceinline__ nnkey_t nn_scan_merged_quad(gpts_ptr pts4, const uint32_t (&first)[NV],
                                                       const uint32_t (&cnt)[NV], uint32_t sub, float qx, float qy, float qz,
                                                       nnkey_t best MH_CARRY_ARG) {
  uint32_t pre[NV + 1], start[NV];
  pre[0] = 0;
#pragma unroll
  for (int v = 0; v < NV; v++) {
    pre[v + 1] = pre[v] + cnt[v];
    start[v] = first[v] - pre[v];
  }
  const uint32_t total = pre[NV];  // the same in the four lanes
  for (uint32_t t0 = 0; t0 < total; t0 += 4 * kQuadW) best = nn_scan_round_quad<NV, kQuadW>(pts4, start, pre, t0, sub, qx, qy, qz, best MH_CARRY_PASS);
  return quad_min_key(best);
}

// The neighbour codes lane `sub` of a quad is responsible for when the bound is evaluated: the nine codes with iz == sub
// (code = ix*9 + iy*3 + iz; lane 3 has none).  Per lane that is one select for the z gap and 4 + 9 additions in the order of
// nn_lower_bound -- (sx + sy) + sz -- instead of seven decompositions of a lane-dependent code (140 vector instructions
// of the 900 a wave spends on its sixteen points went into those).
struct QuadBounds {
  float lb[9];  // conservative lower bounds of codes (ix*3 + iy)*3 + sub, index ix*3 + iy (lane 3: +inf)
};
__device__ __forceinline__ QuadBounds quad_bounds(const Gaps& gx, const Gaps& gy, const Gaps& gz, uint32_t sub) {
  QuadBounds q;
  const float zs = sub == 0 ? gz.s[0] : (sub == 1 ? 0.f : (sub == 2 ? gz.s[2] : __builtin_inff()));
#pragma unroll
  for (int ix = 0; ix < 3; ix++)
#pragma unroll
    for (int iy = 0; iy < 3; iy++) {
      const float sx = ix == 1 ? 0.f : gx.s[ix], sy = iy == 1 ? 0.f : gy.s[iy];
      q.lb[ix * 3 + iy] = ((sx + sy) + zs) * 0.9999f;
    }
  return q;
}
// bit `code` set iff the voxel can still hold the winner; each lane tests its nine codes, the quad ORs them together
__device__ __forceinline__ uint32_t quad_bound_mask(const QuadBounds& q, uint32_t sub, float best) {
  uint32_t mine = 0;
#pragma unroll
  for (int j = 0; j < 9; j++) mine |= (!(q.lb[j] > best)) ? (1u << (3 * j)) : 0u;
  mine = sub < 3u ? mine << sub : 0u;
  mine |= quad_u32<0xB1>(mine);
  mine |= quad_u32<0x4E>(mine);
  return mine & 0x07FFFFFFu & ~(1u << 13);
}

// every lane of the quad passes the same q and gets the same result.
// bound0: an upper bound of the answer's d2 that is KNOWN to be attained by a map record (the distance, in the candidate
// arithmetic, from q to the record this point was paired with in the previous ICP iteration), or +inf.  With a bound the
// search skips the own-voxel-first protocol: every voxel whose lower bound does not exceed it -- the own voxel and, for a
// converging alignment, one neighbour or none -- is probed in ONE round trip and scanned in one more.  The result is the
// same: the answer's voxel has a lower bound <= the answer's d2 <= bound0, so it is scanned, and ties resolve by the
// usual (d2, record) key.  Should bound0 not be attained inside the 27-voxel block (the point moved more than a voxel
// away from its old partner), nothing beats the initial key and the search runs again without a bound.
#ifdef MH_DEBUG_FLOOR
// debug build (tools/match_floor.py): the search writes down WHAT IT TOUCHED for this point -- per probe batch the voxel code of
// each of the quad's lanes -- so that k_match_floor_b can replay the same dependent chain of loads without the arithmetic.
// cap: this point's 2 dwords [0] winner record  [1] batches (4 bits) | codes of the first batch, 6 bits per lane (63 = none);
// cap_more: 4 dwords per point for batches 1..4 (codes, a byte per lane), read by the replay only when there are any
#define MH_FLOOR_ARG , uint32_t* __restrict__ cap = nullptr, uint32_t* __restrict__ cap_more = nullptr
#define MH_FLOOR_BATCH(c_mine_)                                                                                                 \
  do {                                                                                                                          \
    if (cap) {                                                                                                                  \
      const uint32_t mine_ = (uint32_t)((c_mine_) < 0 ? 63 : (c_mine_)) & 63u;                                                  \
      const uint32_t packed_ = quad_bcast<0>(mine_) | (quad_bcast<1>(mine_) << 8) | (quad_bcast<2>(mine_) << 16) | (quad_bcast<3>(mine_) << 24); \
      if (floor_nb == 0u) floor_first = (packed_ & 63u) | (((packed_ >> 8) & 63u) << 6) | (((packed_ >> 16) & 63u) << 12) | (((packed_ >> 24) & 63u) << 18); \
      else if (sub == 0 && floor_nb < 5u) cap_more[floor_nb - 1u] = packed_;                                                    \
      floor_nb++;                                                                                                               \
    }                                                                                                                           \
  } while (0)
#else
#define MH_FLOOR_ARG
#define MH_FLOOR_BATCH(c_mine_) do { } while (0)
#endif
// NARROW (k_match4 built with -DMH_NARROW_IO; round-4 experiment, OFF): the winner's record is fetched ONE DWORD PER LANE --
// lane s of the quad gets component s (x, y, z, source index) in r.pt.x -- instead of the same 16 bytes in all four lanes, and
// the caller reads the previous pairing and writes the new one the same way.  tools/match_floor.py's what-if runs had priced
// the three 16-byte-per-lane accesses around the search at 8.5 % (previous pairing in), 2.5 % (winner's record) and 10 %
// (pairing out) of the launch; narrowing them does not recover that: 0.266-0.277 ms per launch against 0.259 (bit-identical,
// 4370-4540 scans/s against 4590-4620; the replay kernel says the same, 0.2735 against 0.2605 ms) -- what those accesses
// cost is their place in the dependent chain (the first thing a wave waits for, the last thing it has to retire), not their
// width, and the narrow form adds quad_perm moves and 24 bytes of scratch.  profiles/r04_match_kernel.md.
// The sub-voxel index at work: the range [f, f + n) of a probed voxel's records (in pts_q order) narrowed to the hull of the
// quadrants (x half, y half) that can hold a record within the bound `bd` of query (qx, qy).  A record of the low x half has
// x < mid_x, so it is at least q_x - mid_x away from a query right of the mid plane (and likewise for the others); the margins
// and the 0.9999 are the voxel bound's (axis_gaps).  qv: the voxel's packed boundaries (bit 31: it has any); code: the voxel's
// place in the 27-voxel block (-1: none).  Also what tools/match_floor.py's replay kernel does, with the same arguments.
__device__ __forceinline__ void quad_narrow(const MapView& m, uint32_t qv, int code, float qx, float qy, float bd, uint32_t& f,
                                            uint32_t& n) {
  if (!(qv & 0x80000000u) || n == 0u) return;
  const int ix = (code * 57) >> 9, rr = code - 9 * ix, iy = (rr * 11) >> 5;
  // (the own voxel's indices are computed again from opaque copies of the query: kept live from the prologue they cost
  // the kernel two registers it does not have at eight waves per SIMD)
  float qx2 = qx, qy2 = qy;
  asm volatile("" : "+v"(qx2), "+v"(qy2));
  const int vx = voxel_of(qx2, m.inv_vs, 0) - 1 + ix, vy = voxel_of(qy2, m.inv_vs, 0) - 1 + iy;
  const float vs = m.vs;
  const float ex = qx - ((float)vx + 0.5f) * vs, ey = qy - ((float)vy + 0.5f) * vs;
  const float ax_ = fmaxf(0.f, fabsf(ex) - 1.0e-6f * (fabsf((float)vx) + 2.f) * vs);
  const float ay_ = fmaxf(0.f, fabsf(ey) - 1.0e-6f * (fabsf((float)vy) + 2.f) * vs);
  const bool farx = ax_ * ax_ * 0.9999f > bd, fary = ay_ * ay_ * 0.9999f > bd;
  const bool no_xlo = farx && ex > 0.f, no_xhi = farx && ex < 0.f, no_ylo = fary && ey > 0.f, no_yhi = fary && ey < 0.f;
  const uint32_t lo_q = (no_xlo ? 2u : 0u) + (no_ylo ? 1u : 0u);   // first needed quadrant (2 * xhalf + yhalf)
  const uint32_t hi_q = (no_xhi ? 0u : 2u) + (no_yhi ? 0u : 1u);   // last needed quadrant
  const uint32_t b_lo = lo_q == 0u ? 0u : ((qv >> (8u + 5u * (lo_q - 1u))) & 31u);
  const uint32_t b_hi = hi_q == 3u ? n : ((qv >> (8u + 5u * hi_q)) & 31u);
  f += b_lo;
  n = b_hi - b_lo;
}

template <bool NARROW = false>
__device__ __forceinline__ NNResult nn_search_quad(const MapView& m, uint32_t sub, float qx, float qy, float qz,
                                                   float bound0 = __builtin_inff() MH_FLOOR_ARG) {
  NNResult r;
#ifdef MH_DEBUG_FLOOR
  uint32_t floor_nb = 0, floor_first = 0x00FFFFFFu;
#endif
  r.d2 = __builtin_inff();
  r.found = false;
  r.pt = (f32x4)(0.f);
  const float lim = 1.0e6f;  // one test, no short-circuit branches: NaN and inf fail it as well
  if (!((int)(fabsf(qx * m.inv_vs) < lim) & (int)(fabsf(qy * m.inv_vs) < lim) & (int)(fabsf(qz * m.inv_vs) < lim))) return r;
  const int cx = voxel_of(qx, m.inv_vs, m.trunc), cy = voxel_of(qy, m.inv_vs, m.trunc), cz = voxel_of(qz, m.inv_vs, m.trunc);
  const unsigned long long kbase = pack_key(cx - 1, cy - 1, cz - 1);
  const gslots_ptr slots4 = (gslots_ptr)m.slots;
  const gpts_ptr pts4 = (gpts_ptr)m.pts;
  const gpts_ptr spts = (gpts_ptr)m.pts_q;  // what the scans read (map_ensure_qidx has run); the winner is fetched from pts
  const float vs = m.vs;
  const Gaps gx = axis_gaps(qx, cx, vs, m.trunc), gy = axis_gaps(qy, cy, vs, m.trunc), gz = axis_gaps(qz, cz, vs, m.trunc);
  const QuadBounds qb = quad_bounds(gx, gy, gz, sub);
  nnkey_t best = kNNKeyNone;
#ifdef MH_CARRY_WINNER
  f32x4 brec = (f32x4)(0.f);
  nnkey_t mine = kNNKeyNone;  // this lane's own best (d2, record) and, in brec, that record
#endif
#ifdef MH_DEBUG_WAVETRACE
  if (m.dbg_stop == 1) return r;  // prologue only
#endif
  const uint32_t kFaces = (1u << 4) | (1u << 10) | (1u << 12) | (1u << 14) | (1u << 16) | (1u << 22);
  const uint32_t kCorners = (1u << 0) | (1u << 2) | (1u << 6) | (1u << 8) | (1u << 18) | (1u << 20) | (1u << 24) | (1u << 26);
  const uint32_t kEdges = 0x07FFFFFFu & ~(kFaces | kCorners | (1u << 13));
  const bool bounded = bound0 < __builtin_inff();  // the same in the four lanes
  uint32_t todo = 0x07FFFFFFu;
  if (bounded) {
    best = ((nnkey_t)__float_as_uint(bound0) << 32) | 0xFFFFFFFFull;
  } else {
    // Speculative probes, issued together with the own-voxel probe (same round trip): the three faces on the near side
    // of each axis and the edge between the two nearest of them -- almost always the only neighbours that survive the
    // bound once the own voxel has been scanned.  Lane s of the quad probes the s-th of them.
    const int sx = gx.s[0] <= gx.s[2] ? 0 : 2, sy = gy.s[0] <= gy.s[2] ? 0 : 2, sz = gz.s[0] <= gz.s[2] ? 0 : 2;
    const float nx = gx.s[sx], ny = gy.s[sy], nz = gz.s[sz];
    const int c_fx = sx * 9 + 3 + 1, c_fy = 9 + sy * 3 + 1, c_fz = 9 + 3 + sz;
    const int c_e = (nz >= nx && nz >= ny) ? sx * 9 + sy * 3 + 1 : (ny >= nx ? sx * 9 + 3 + sz : 9 + sy * 3 + sz);
    const int c_spec = sub == 0 ? c_fx : (sub == 1 ? c_fy : (sub == 2 ? c_fz : c_e));
    const uint32_t spec_bits = (1u << c_fx) | (1u << c_fy) | (1u << c_fz) | (1u << c_e);
    const unsigned long long key_s = nn_key_of(kbase, c_spec);
    // the query's own voxel (code 13): the four lanes read the same slot
    const unsigned long long key = nn_key_of(kbase, 13);
    const u32x4 sl_c = slots4[hash_key(key) & m.mask];
    const u32x4 sl_s = slots4[hash_key(key_s) & m.mask];
    uint32_t f1[1], c1[1];
    nn_resolve(m, slots4, key, sl_c, true, f1[0], c1[0]);
#ifdef MH_DEBUG_WAVETRACE
    if (m.dbg_stop == 2) { r.d2 = (float)(f1[0] + c1[0] + sl_s.z); return r; }  // + own-voxel probe
#endif
    best = nn_scan_merged_quad<1>(spts, f1, c1, sub, qx, qy, qz, best MH_CARRY_PASS);
#ifdef MH_DEBUG_WAVETRACE
    if (m.dbg_stop == 3) { r.d2 = nnkey_d2(best) + (float)sl_s.z; return r; }  // + own-voxel scan
#endif
    // the speculated neighbours that did survive: one merged scan, no further probe round trip
    const uint32_t live0 = quad_bound_mask(qb, sub, nnkey_d2(best));
    uint32_t f_s, n_s;
    nn_resolve(m, slots4, key_s, sl_s, ((live0 >> c_spec) & 1u) != 0, f_s, n_s);
    // (the edge lane may repeat a face code when two gaps tie at zero: count it once)
    if (sub == 3 && (c_e == c_fx || c_e == c_fy || c_e == c_fz)) n_s = 0;
    const uint32_t first[4] = {quad_bcast<0>(f_s), quad_bcast<1>(f_s), quad_bcast<2>(f_s), quad_bcast<3>(f_s)};
    const uint32_t cnt[4] = {quad_bcast<0>(n_s), quad_bcast<1>(n_s), quad_bcast<2>(n_s), quad_bcast<3>(n_s)};
    best = nn_scan_merged_quad<4>(spts, first, cnt, sub, qx, qy, qz, best MH_CARRY_PASS);
    todo &= ~(1u << 13) & ~spec_bits;
  }
  for (int pass = 0; pass < 2; pass++) {
    // voxels that can still hold the winner, the own voxel always among them while it is to do.  The bound only ever
    // tightens, so the set only shrinks: it is re-evaluated after a batch only if something of it is left.
    uint32_t cand = todo & (quad_bound_mask(qb, sub, nnkey_d2(best)) | (1u << 13));
    while (cand) {
      int c_mine = -1;
      if (bounded && pass == 0) {
        // with a bound: lane 0 takes the own voxel, the others the three lowest codes of the rest (the order matters
        // little when the bound is tight already: nearly always this is the only batch)
        const uint32_t own = cand & (1u << 13);
        uint32_t rem = cand & ~(1u << 13);
        const uint32_t b1 = rem & (0u - rem);
        rem ^= b1;
        const uint32_t b2 = rem & (0u - rem);
        rem ^= b2;
        const uint32_t b3 = rem & (0u - rem);
        const uint32_t bm = sub == 0 ? own : (sub == 1 ? b1 : (sub == 2 ? b2 : b3));
        c_mine = bm ? __builtin_ctz(bm) : -1;
        cand &= ~(own | b1 | b2 | b3);
      } else {
        // without one: four of them from the nearest class that has any -- own voxel and faces, then edges, then
        // corners (the scan of the faces usually prunes the edges)
        const uint32_t ln = cand & (kFaces | (1u << 13)), le = cand & kEdges;
        uint32_t rem = ln ? ln : (le ? le : cand);
#pragma unroll
        for (int v = 0; v < 4; v++) {
          const int cv = rem ? __builtin_ctz(rem) : -1;
          rem &= rem - 1;
          if (cv >= 0) cand &= ~(1u << cv);
          c_mine = (uint32_t)v == sub ? cv : c_mine;
        }
      }
      MH_FLOOR_BATCH(c_mine);
      const unsigned long long key = nn_key_of(kbase, c_mine < 0 ? 0 : c_mine);
      const u32x4 sl = slots4[hash_key(key) & m.mask];  // one probe per lane, four per point in flight
      // the quadrant boundaries of the probed voxel come in its slot's count word, wherever linear probing finds the slot
      uint32_t f_mine, n_mine, qv;
      nn_resolve(m, slots4, key, sl, c_mine >= 0, f_mine, n_mine, &qv);
      quad_narrow(m, qv, c_mine, qx, qy, nnkey_d2(best), f_mine, n_mine);
      const uint32_t first[4] = {quad_bcast<0>(f_mine), quad_bcast<1>(f_mine), quad_bcast<2>(f_mine), quad_bcast<3>(f_mine)};
      const uint32_t cnt[4] = {quad_bcast<0>(n_mine), quad_bcast<1>(n_mine), quad_bcast<2>(n_mine), quad_bcast<3>(n_mine)};
      best = nn_scan_merged_quad<4>(spts, first, cnt, sub, qx, qy, qz, best MH_CARRY_PASS);
      if (cand) cand &= quad_bound_mask(qb, sub, nnkey_d2(best)) | (1u << 13);
    }
    if (!bounded || nnkey_idx(best) != 0xFFFFFFFFu) break;
    // the bound was not attained inside the block: once more, without it
    best = kNNKeyNone;
    todo = 0x07FFFFFFu;
#ifdef MH_CARRY_WINNER
    mine = kNNKeyNone;
#endif
  }
#ifdef MH_DEBUG_WAVETRACE
  if (m.dbg_stop == 4) { r.d2 = nnkey_d2(best); return r; }  // + neighbours, without the final record fetch
#endif
#ifdef MH_DEBUG_FLOOR
  if (cap && sub == 0) {
    cap[0] = nnkey_idx(best);
    // (the un-bounded first iteration has a prologue of its own: not replayed -> 15 batches = "not captured")
    cap[1] = ((bounded ? (floor_nb < 14u ? floor_nb : 14u) : 15u) << 24) | floor_first;
  }
#endif
#ifdef MH_CARRY_WINNER
  // keys are unique (the record index is part of them): exactly one lane of the quad holds the winner as ITS best.
  // r.writer says whether this lane is the one (the caller lets that lane store the pairing instead of lane 0).
  r.writer = mine == best;
  if (nnkey_idx(best) != 0xFFFFFFFFu) {
    r.pt = brec;
    r.d2 = nnkey_d2(best);
    r.found = true;
  } else {
    r.writer = sub == 0;  // nothing found: lane 0 reports it
  }
  return r;
#else
  if (nnkey_idx(best) != 0xFFFFFFFFu) {
    if (NARROW) r.pt.x = reinterpret_cast<const float MH_AS_GLOBAL*>(pts4)[4ull * nnkey_idx(best) + sub];
    else r.pt = pts4[nnkey_idx(best)];
    r.d2 = nnkey_d2(best);
    r.found = true;
  }
  return r;
#endif
}

template <int CTRL>
__device__ __forceinline__ uint32_t row_dpp_keep(uint32_t old, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t row_bcast_u32(uint32_t v, uint32_t src_lane_in_row) {
  const uint32_t lane = (uint32_t)__lane_id();
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane & ~15u) + src_lane_in_row) << 2), (int)v);
}
// min over the 16 lanes of a row, valid in every lane of the row afterwards
__device__ __forceinline__ nnkey_t row_min_key(nnkey_t k) {
#define MH_ROW_STEP(CTRL)                                                                                         \
  {                                                                                                               \
    const nnkey_t o = ((nnkey_t)row_dpp_keep<CTRL>(0xFFFFFFFFu, (uint32_t)(k >> 32)) << 32) |                     \
                      row_dpp_keep<CTRL>(0xFFFFFFFFu, (uint32_t)k);                                               \
    k = o < k ? o : k;                                                                                            \
  }
  MH_ROW_STEP(0x101)  // row_shl:1 (lane i reads lane i+1 of its row; out-of-row lanes keep `old` = all ones)
  MH_ROW_STEP(0x102)
  MH_ROW_STEP(0x104)
  MH_ROW_STEP(0x108)  // lane 0 of the row holds the minimum
#undef MH_ROW_STEP
  return ((nnkey_t)row_bcast_u32((uint32_t)(k >> 32), 0) << 32) | row_bcast_u32((uint32_t)k, 0);
}

constexpr int kRowW = 2;  // records per lane and round trip: 16 x 2 = 32 >= one full voxel (cap 20)

template <int NV>
__device__ __forceinline__ nnkey_t nn_scan_merged_row(gpts_ptr pts4, const uint32_t (&first)[NV],
                                                      const uint32_t (&cnt)[NV], uint32_t r16, float qx, float qy, float qz,
                                                      nnkey_t best) {
  uint32_t pre[NV + 1], start[NV];
  pre[0] = 0;
#pragma unroll
  for (int v = 0; v < NV; v++) {
    pre[v + 1] = pre[v] + cnt[v];
    start[v] = first[v] - pre[v];
  }
  const uint32_t total = pre[NV];  // the same in the sixteen lanes
  for (uint32_t t0 = 0; t0 < total; t0 += 16 * kRowW) {
    f32x4 c[kRowW];
    uint32_t ri[kRowW];
    bool valid[kRowW];
#pragma unroll
    for (int u = 0; u < kRowW; u++) {
      const uint32_t tu = t0 + 16u * (uint32_t)u + r16;
      valid[u] = tu < total;
      const uint32_t t = valid[u] ? tu : total - 1;
      uint32_t off = start[0];
#pragma unroll
      for (int v = 1; v < NV; v++) off = t >= pre[v] ? start[v] : off;
      ri[u] = t + off;
      c[u] = pts4[ri[u]];
    }
#pragma unroll
    for (int u = 0; u < kRowW; u++) {
      const float dx = c[u].x - qx, dy = c[u].y - qy, dz = c[u].z - qz;
      const float d2 = (dx * dx + dy * dy) + dz * dz;  // fp32, un-fused, this order (bit-exact with the oracle)
      const nnkey_t k = valid[u] ? (((nnkey_t)__float_as_uint(d2) << 32) | ri[u]) : kNNKeyNone;
      best = k < best ? k : best;
    }
  }
  return row_min_key(best);
}

// every lane of the row passes the same q and gets the same result
// bound0: as for nn_search_quad -- the distance to the record paired with this point in the previous iteration (or +inf):
// the own voxel then joins the first batch of neighbours instead of having a round trip of its own.
__device__ __forceinline__ NNResult nn_search_row16(const MapView& m, uint32_t r16, float qx, float qy, float qz,
                                                    float bound0 = __builtin_inff()) {
  NNResult r;
  r.d2 = __builtin_inff();
  r.found = false;
  r.pt = (f32x4)(0.f);
  const float lim = 1.0e6f;  // one test, no short-circuit branches: NaN and inf fail it as well
  if (!((int)(fabsf(qx * m.inv_vs) < lim) & (int)(fabsf(qy * m.inv_vs) < lim) & (int)(fabsf(qz * m.inv_vs) < lim))) return r;
  const int cx = voxel_of(qx, m.inv_vs, m.trunc), cy = voxel_of(qy, m.inv_vs, m.trunc), cz = voxel_of(qz, m.inv_vs, m.trunc);
  const unsigned long long kbase = pack_key(cx - 1, cy - 1, cz - 1);
  const gslots_ptr slots4 = (gslots_ptr)m.slots;
  const gpts_ptr pts4 = (gpts_ptr)m.pts;
  const float vs = m.vs;
  const Gaps gx = axis_gaps(qx, cx, vs, m.trunc), gy = axis_gaps(qy, cy, vs, m.trunc), gz = axis_gaps(qz, cz, vs, m.trunc);
  // round 1: lane r owns codes r ("a") and r + 16 ("b", only r <= 10)
  const int ca = (int)r16, cb = (int)r16 + 16;
  const bool has_b = cb < 27;
  const unsigned long long ka = nn_key_of(kbase, ca), kb = nn_key_of(kbase, has_b ? cb : ca);
  const u32x4 sa = slots4[hash_key(ka) & m.mask];
  const u32x4 sb = slots4[hash_key(kb) & m.mask];
  uint32_t fa, na, fb, nb;
  nn_resolve(m, slots4, ka, sa, true, fa, na);
  nn_resolve(m, slots4, kb, sb, has_b, fb, nb);
  const bool bounded = bound0 < __builtin_inff();  // the same in the sixteen lanes
  const float lba = ca == 13 ? 0.f : nn_lower_bound(ca, gx, gy, gz) * 0.9999f;  // (the own voxel is always live)
  const float lbb = has_b ? nn_lower_bound(cb, gx, gy, gz) * 0.9999f : __builtin_inff();
  nnkey_t best = bounded ? (((nnkey_t)__float_as_uint(bound0) << 32) | 0xFFFFFFFFull) : kNNKeyNone;
  const uint32_t row_shift = (uint32_t)__lane_id() & 48u;
  for (int pass = 0; pass < 2; pass++) {
    const bool own_first = !bounded || pass == 1;
    if (own_first) {  // round 2: the own voxel (code 13, slot "a" of lane 13)
      const uint32_t f1[1] = {row_bcast_u32(fa, 13)}, c1[1] = {row_bcast_u32(na, 13)};
      best = nn_scan_merged_row<1>(pts4, f1, c1, r16, qx, qy, qz, best);
    }
    // rounds 3+: surviving non-empty neighbours, four at a time, lowest code first (any order gives the same minimum)
    bool todo_a = (ca != 13 || !own_first) && na > 0, todo_b = has_b && nb > 0;
    for (;;) {
      const float bd = nnkey_d2(best);
      const bool pa = todo_a && !(lba > bd), pb = todo_b && !(lbb > bd);
      const uint32_t ma = (uint32_t)(__ballot(pa) >> row_shift) & 0xFFFFu, mb = (uint32_t)(__ballot(pb) >> row_shift) & 0xFFFFu;
      uint32_t mm = ma | (mb << 16);  // bit = code
      if (!mm) break;
      uint32_t first[4], cnt[4];
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int code = mm ? __builtin_ctz(mm) : -1;
        mm &= mm - 1;
        const uint32_t owner = (uint32_t)(code < 0 ? 0 : code) & 15u;
        const bool is_b = code >= 16;
        const uint32_t f = row_bcast_u32(is_b ? fb : fa, owner), n = row_bcast_u32(is_b ? nb : na, owner);
        // NB: the value is selected BEFORE the permute in every lane, so the owner lane contributes the right slot
        first[v] = f;
        cnt[v] = code < 0 ? 0u : n;
        if (code >= 0) {
          if (code == ca) todo_a = false;
          if (code == cb) todo_b = false;
        }
      }
      best = nn_scan_merged_row<4>(pts4, first, cnt, r16, qx, qy, qz, best);
    }
    if (!bounded || nnkey_idx(best) != 0xFFFFFFFFu) break;
    best = kNNKeyNone;  // the bound was not attained inside the block: once more, without it
  }
  if (nnkey_idx(best) != 0xFFFFFFFFu) {
    r.pt = pts4[nnkey_idx(best)];
    r.d2 = nnkey_d2(best);
    r.found = true;
  }
  return r;
}

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

template <int CTRL>
__device__ __forceinline__ int dpp_keep_i32(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xF, 0xF, false);  // lanes without a source keep `old`
}
__device__ __forceinline__ int wave_min_i32(int v) {
  const int big = 0x7FFFFFFF;
  v = min(v, dpp_keep_i32<0x101>(big, v));  // row_shl:1
  v = min(v, dpp_keep_i32<0x102>(big, v));
  v = min(v, dpp_keep_i32<0x104>(big, v));
  v = min(v, dpp_keep_i32<0x108>(big, v));  // lane 0 of each row holds the row's minimum
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max_i32(int v) { return -wave_min_i32(-v); }  // (callers stay away from INT_MIN)

// inclusive prefix sum over the 64 lanes of a wave (DPP row shifts inside the 16-lane rows, three readlanes across)
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);  // row_shr:1 (lane i reads lane i-1 of its row, else 0)
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
  const int t0 = __builtin_amdgcn_readlane(x, 15), t1 = __builtin_amdgcn_readlane(x, 31), t2 = __builtin_amdgcn_readlane(x, 47);
  const uint32_t row = ((uint32_t)__lane_id()) >> 4;
  x += (row >= 1 ? t0 : 0) + (row >= 2 ? t1 : 0) + (row >= 3 ? t2 : 0);
  return (uint32_t)x;
}

// all records of one voxel's run in LDS, every lane its own run (cnt may be 0), four reads in flight per lane
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
__device__ __forceinline__ void wave_sync_lds_nn() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
constexpr uint32_t kWaveMinPoints = 40;
constexpr uint32_t kWaveMaxVox = 128;

__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, uint32_t lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane);
}

// A pointer in the CONSTANT address space built from wave-uniform bits: loads through it with a wave-uniform index are
// scalar loads (s_load_dwordx4/x8 into SGPRs, the scalar data cache) -- the map is read-only for the whole launch.
typedef const f32x4 __attribute__((address_space(4))) * cf32x4_ptr;
__device__ __forceinline__ cf32x4_ptr uniform_const_ptr(const void* p) {
  const unsigned long long a = (unsigned long long)p;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  return (cf32x4_ptr)(((unsigned long long)hi << 32) | lo);
}

// Four records [f + j, f + j + 4) of a run of c, indices clamped into the run (a duplicate of the last record can neither
// lower a minimum nor win a strict '<'): four scalar loads, wave-uniform addresses.
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
  // fp64 moments held to a relative tolerance, not bit-compared: let the multiply-adds fuse here (126 -> ~75
  // instructions per point); everything that decides WHICH points pair stays un-fused (file header)
#pragma clang fp contract(fast)
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

// The same without a branch: an unpaired (or non-finite) point is accumulated with weight 0 on sanitised inputs, which
// leaves every sum unchanged.  Straight-line code lets the compiler interleave the dependent fp64 chains of the
// several points a lane handles (under `if (paired)` each point's chain ran on its own).
__device__ __forceinline__ void acc_pt2pt_masked(Acc& a, const double* __restrict__ T, bool paired, float lxf, float lyf,
                                                 float lzf, float qxf, float qyf, float qzf, uint32_t kernel, double kparam,
                                                 double wpair) {
  acc_pt2pt(a, T, paired ? lxf : 0.f, paired ? lyf : 0.f, paired ? lzf : 0.f, paired ? qxf : 0.f, paired ? qyf : 0.f,
            paired ? qzf : 0.f, kernel, kparam, paired ? wpair : 0.0);
  a.v[17] -= paired ? 0.0 : 1.0;  // acc_pt2pt counted it
}

// Sum over the 64 lanes of a wave, result valid in every lane... of interest only in lane 0.
// DPP row shifts (VALU, no LDS traffic) reduce each 16-lane row to its lane 0, then the four row heads are read
// with v_readlane and added in a fixed order.  (The first version used 6 x __shfl_xor = 12 ds_bpermute per value;
// with 18 fp64 moments per wave that was ~200 dependent LDS round trips at the tail of every match/accum launch.)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const unsigned long long b = __double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xF, 0xF, true);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, true);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const unsigned long long b = __double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), l);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0x101>(v);  // row_shl:1  (lane i += lane i+1 of its row; out-of-row reads 0)
  v += dpp_f64<0x102>(v);  // row_shl:2
  v += dpp_f64<0x104>(v);  // row_shl:4
  v += dpp_f64<0x108>(v);  // row_shl:8  -> lane 0 of each row holds the row sum
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
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
