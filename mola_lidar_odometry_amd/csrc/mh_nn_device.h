// mh_nn_device.h -- device-side building blocks of the correspondence search and of the Gauss-Newton
// accumulation.  Everything that decides WHICH point pairs (transform, voxel index, fp32 distance,
// comparison order) is written un-fused and in a fixed order so that the result is bit-identical to
// the CPU restatement of the reference algorithm (this file is compiled with -ffp-contract=off).
#pragma once
#include "mh_internal.h"

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
template <int NV, int W>
__device__ __forceinline__ nnkey_t nn_scan_round_quad(gpts_ptr pts4, const uint32_t (&start)[NV],
                                                      const uint32_t (&pre)[NV + 1], uint32_t t0, uint32_t sub, float qx,
                                                      float qy, float qz, nnkey_t best) {
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
    best = k < best ? k : best;
  }
  return best;
}

template <int NV>
__device__ __forceinline__ nnkey_t nn_scan_merged_quad(gpts_ptr pts4, const uint32_t (&first)[NV],
                                                       const uint32_t (&cnt)[NV], uint32_t sub, float qx, float qy, float qz,
                                                       nnkey_t best) {
  uint32_t pre[NV + 1], start[NV];
  pre[0] = 0;
#pragma unroll
  for (int v = 0; v < NV; v++) {
    pre[v + 1] = pre[v] + cnt[v];
    start[v] = first[v] - pre[v];
  }
  const uint32_t total = pre[NV];  // the same in the four lanes
  for (uint32_t t0 = 0; t0 < total; t0 += 4 * kQuadW) best = nn_scan_round_quad<NV, kQuadW>(pts4, start, pre, t0, sub, qx, qy, qz, best);
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
// (Round 4 measured the three 16-byte accesses around the search -- previous pairing in, winner's record, pairing out -- a dword
// per lane instead: bit-identical and slower, 0.266-0.277 ms per launch against 0.259; profiles/r04_match_kernel.md.  Removed.)
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

__device__ __forceinline__ NNResult nn_search_quad(const MapView& m, uint32_t sub, float qx, float qy, float qz,
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
  const gpts_ptr spts = (gpts_ptr)m.pts_q;  // what the scans read (map_ensure_qidx has run); the winner is fetched from pts
  const float vs = m.vs;
  const Gaps gx = axis_gaps(qx, cx, vs, m.trunc), gy = axis_gaps(qy, cy, vs, m.trunc), gz = axis_gaps(qz, cz, vs, m.trunc);
  const QuadBounds qb = quad_bounds(gx, gy, gz, sub);
  nnkey_t best = kNNKeyNone;
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
    best = nn_scan_merged_quad<1>(spts, f1, c1, sub, qx, qy, qz, best);
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
    best = nn_scan_merged_quad<4>(spts, first, cnt, sub, qx, qy, qz, best);
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
      const unsigned long long key = nn_key_of(kbase, c_mine < 0 ? 0 : c_mine);
      const u32x4 sl = slots4[hash_key(key) & m.mask];  // one probe per lane, four per point in flight
      // the quadrant boundaries of the probed voxel come in its slot's count word, wherever linear probing finds the slot
      uint32_t f_mine, n_mine, qv;
      nn_resolve(m, slots4, key, sl, c_mine >= 0, f_mine, n_mine, &qv);
      quad_narrow(m, qv, c_mine, qx, qy, nnkey_d2(best), f_mine, n_mine);
      const uint32_t first[4] = {quad_bcast<0>(f_mine), quad_bcast<1>(f_mine), quad_bcast<2>(f_mine), quad_bcast<3>(f_mine)};
      const uint32_t cnt[4] = {quad_bcast<0>(n_mine), quad_bcast<1>(n_mine), quad_bcast<2>(n_mine), quad_bcast<3>(n_mine)};
      best = nn_scan_merged_quad<4>(spts, first, cnt, sub, qx, qy, qz, best);
      if (cand) cand &= quad_bound_mask(qb, sub, nnkey_d2(best)) | (1u << 13);
    }
    if (!bounded || nnkey_idx(best) != 0xFFFFFFFFu) break;
    // the bound was not attained inside the block: once more, without it
    best = kNNKeyNone;
    todo = 0x07FFFFFFu;
  }
#ifdef MH_DEBUG_WAVETRACE
  if (m.dbg_stop == 4) { r.d2 = nnkey_d2(best); return r; }  // + neighbours, without the final record fetch
#endif
  if (nnkey_idx(best) != 0xFFFFFFFFu) {
    r.pt = pts4[nnkey_idx(best)];
    r.d2 = nnkey_d2(best);
    r.found = true;
  }
  return r;
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

// (generic wave helpers; the tile / wave searches that came with them live in mh_nn_dev_variants.h)
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
__device__ __forceinline__ void wave_sync_lds_nn() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
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
#ifdef MH_DEV_VARIANTS
#include "mh_nn_dev_variants.h"  // nn_search_tile, nn_search_wave
#endif

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
