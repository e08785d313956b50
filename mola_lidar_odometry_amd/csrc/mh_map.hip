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

#include <algorithm>
#include <new>
#include <vector>

#include "mh_internal.h"

using namespace mh;

namespace {

__device__ __forceinline__ int voxel_index(float c, float inv_vs, uint32_t trunc) {
  const float s = c * inv_vs;  // fp32 product, like the reference
  return trunc ? (int)s : (int)floorf(s);
}

// remove_voxels_farther_than's voxel-index distance test (mh_map_params::far_voxel_metric, MH_FAR_*)
__device__ __forceinline__ bool far_voxel(int dx, int dy, int dz, int dist, uint32_t metric) {
  dx = abs(dx); dy = abs(dy); dz = abs(dz);
  if (metric == MH_FAR_L1) return (long long)dx + dy + dz > (long long)dist;
  if (metric == MH_FAR_L2) return (long long)dx * dx + (long long)dy * dy + (long long)dz * dz > (long long)dist * dist;
  return max(max(dx, dy), dz) > dist;
}

__global__ void k_keys(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, uint32_t n,
                       float inv_vs, uint32_t trunc, int4 evict /* {cx,cy,cz,dist_in_grid}; w < 0 = off */, uint32_t metric,
                       unsigned long long* __restrict__ keys, uint32_t* __restrict__ idx, uint32_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float px = x[i], py = y[i], pz = z[i];
  unsigned long long k = kEmptyKey;
  if (isfinite(px) && isfinite(py) && isfinite(pz)) {
    const float sx = px * inv_vs, sy = py * inv_vs, sz = pz * inv_vs;
    if (fabsf(sx) < 1.0e6f && fabsf(sy) < 1.0e6f && fabsf(sz) < 1.0e6f) {
      const int kx = voxel_index(px, inv_vs, trunc), ky = voxel_index(py, inv_vs, trunc), kz = voxel_index(pz, inv_vs, trunc);
      // remove_voxels_farther_than (yaml:238): erasing a voxel after the insertion == never storing its points
      const bool far = evict.w >= 0 && far_voxel(kx - evict.x, ky - evict.y, kz - evict.z, evict.w, metric);
      if (!far) k = pack_key(kx, ky, kz);
    } else {
      atomicOr(&flags[0], 1u);  // voxel index does not fit 21 bits
    }
  }
  keys[i] = k;
  idx[i] = i;
}

// head[i] = 1 where a new voxel run starts; counters[1] = number of valid (finite) points.
// evict.w >= 0 (merge path of mh_map_insert): remove_voxels_farther_than applied to the SORTED keys on the way (k_keys
// left them alone so that the stored points stay in order) -- a far voxel's keys become empty, and the runs of empty
// keys this leaves inside the sequence are skipped by everything below.  A neighbour's key may or may not have been
// rewritten by its own thread yet: the test is repeated on whatever is read, so both readings agree.
__global__ void k_heads(unsigned long long* __restrict__ ks, uint32_t n, uint32_t* __restrict__ head,
                        uint32_t* __restrict__ counters, int4 evict, uint32_t metric) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  auto live = [&](unsigned long long k) {
    if (k == kEmptyKey) return false;
    if (evict.w < 0) return true;
    int kx, ky, kz;
    unpack_key(k, kx, ky, kz);
    return !far_voxel(kx - evict.x, ky - evict.y, kz - evict.z, evict.w, metric);
  };
  const unsigned long long k = ks[i];
  const bool valid = live(k);
  if (!valid && k != kEmptyKey) ks[i] = kEmptyKey;
  // (equal keys share their fate, so comparing with the neighbour's ORIGINAL or rewritten key gives the same head flag)
  head[i] = (valid && (i == 0 || ks[i - 1] != k)) ? 1u : 0u;
  // (runs of evicted keys INSIDE the sequence: several valid-to-empty boundaries -> the maximum)
  if (valid && (i + 1 == n || !live(ks[i + 1]))) atomicMax(&counters[1], i + 1);
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

__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, uint32_t lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane);
}
__device__ __forceinline__ float readlane_f32(float v, uint32_t lane) {
  return __uint_as_float(readlane_u32(__float_as_uint(v), lane));
}
// LDS hand-off between lanes of ONE wave: its LDS operations execute in order, only the compiler has to be kept from
// moving the accesses across this point
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// insertPoint with min_distance_between_points > 0 is order dependent inside a voxel (a point is dropped when it is
// closer than that to an ALREADY STORED point of its voxel, lidar3d-ndt.yaml:244).  Round 4: a WAVE walks the new part of a
// voxel run (one thread walked the whole run, reading its own verdicts back from memory: 125 us per key-frame of the NDT
// pipeline, the longest run's chain of dependent loads).  Stored points (inputs [0, n_stored): they passed the test and the
// cap when they were inserted, and sort first in their run) are kept without a walk; the wave that holds a run's FIRST NEW
// entry copies the run's stored points into its part of LDS and takes the new entries 64 at a time: a candidate is compared
// with all accepted points at once (lane q: accepted point q, q + 64, ...), one ballot decides, the order of the decisions is
// the run's order -- the verdicts are those of the sequential walk.  A run that accepts more than `lds_points` points goes
// on testing the overflow in memory (stored ones by position, new ones through the verdicts: a fence + agent-scope loads,
// other lanes of this wave wrote them).
constexpr uint32_t kKeepLds = 448;     // accepted points per wave held in LDS (3 floats each; 4 waves: 21 KiB)
constexpr uint32_t kKeepBlock = 256;   // k_keep_seq's block size (the LDS array is sized for it)
__global__ __launch_bounds__(kKeepBlock) void k_keep_seq(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ z, const unsigned long long* __restrict__ ks,
                                                         const uint32_t* __restrict__ idx_s, const uint32_t* __restrict__ head,
                                                         const uint32_t* __restrict__ vid1, const uint32_t* __restrict__ vstart,
                                                         uint32_t n, uint32_t cap, float min_dist,
                                                         uint32_t n_stored /* inputs [0, n_stored) are the map's stored points */,
                                                         uint32_t lds_points /* <= kKeepLds (smaller: tests of the overflow path) */,
                                                         uint32_t* __restrict__ keep) {
  __shared__ float acc[kKeepBlock / 64][3][kKeepLds];
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < n && ks[i] != kEmptyKey;
  const uint32_t si = valid ? idx_s[i] : 0u;
  const bool is_new = valid && si >= n_stored;
  if (i < n && !is_new) keep[i] = valid ? 1u : 0u;
  // the first new entry of a run: a head, or the entry behind a stored one (same key: only heads follow another key)
  const bool starter = is_new && (head[i] != 0 || idx_s[i - 1] < n_stored);
  unsigned long long starters = __ballot(starter);
  const float md2 = min_dist * min_dist;
  float* const ax = acc[w][0];
  float* const ay = acc[w][1];
  float* const az = acc[w][2];
  while (starters) {
    const uint32_t st = (i - lane) + (uint32_t)__builtin_ctzll(starters);  // (wave-uniform)
    starters &= starters - 1;
    const unsigned long long k = ks[st];
    const uint32_t a = vstart[vid1[st] - 1u];  // first entry of the run; [a, st) are its stored points
    const uint32_t n_st = st - a;
    wave_sync_lds();  // (the previous run's readers are done with the list)
    for (uint32_t q = lane; q < n_st && q < lds_points; q += 64) {
      const uint32_t sq = idx_s[a + q];
      ax[q] = x[sq]; ay[q] = y[sq]; az[q] = z[sq];
    }
    wave_sync_lds();
    uint32_t kept = n_st;  // wave-uniform
    for (uint32_t base = st;; base += 64) {
      const uint32_t j = base + lane;
      const bool in = j < n && ks[j] == k;
      const uint32_t cnt = (uint32_t)__builtin_popcountll(__ballot(in));  // (a prefix of the lanes: equal keys are contiguous)
      const uint32_t sj = in ? idx_s[j] : 0u;
      const float px = in ? x[sj] : 0.f, py = in ? y[sj] : 0.f, pz = in ? z[sj] : 0.f;
      uint32_t my_ok = 0;
      for (uint32_t c = 0; c < cnt; c++) {
        bool ok = (cap == 0 || kept < cap);  // wave-uniform
        const float cx = readlane_f32(px, c), cy = readlane_f32(py, c), cz = readlane_f32(pz, c);
        if (ok) {
          bool close = false;
          const uint32_t in_lds = kept < lds_points ? kept : lds_points;
          for (uint32_t q = lane; q < in_lds; q += 64) {
            const float dx = ax[q] - cx, dy = ay[q] - cy, dz = az[q] - cz;
            close = close || (((dx * dx + dy * dy) + dz * dz) < md2);
          }
          if (kept > lds_points) {
            // overflow: accepted points number lds_points and up -- stored ones by position, new ones of earlier chunks
            // through their verdicts in memory, this chunk's in registers (my_ok of lanes < c)
            for (uint32_t q = lds_points + lane; q < n_st; q += 64) {
              const uint32_t sq = idx_s[a + q];
              const float dx = x[sq] - cx, dy = y[sq] - cy, dz = z[sq] - cz;
              close = close || (((dx * dx + dy * dy) + dz * dz) < md2);
            }
            __threadfence();
            uint32_t seen = n_st;  // accepted points passed so far
            for (uint32_t qb = st; qb < base; qb += 64) {
              const uint32_t q = qb + lane;
              const uint32_t kq = __hip_atomic_load(&keep[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              const unsigned long long km = __ballot(kq != 0);
              const uint32_t before = seen + (uint32_t)__builtin_popcountll(km & ((1ull << lane) - 1ull));
              if (kq && before >= lds_points) {
                const uint32_t sq = idx_s[q];
                const float dx = x[sq] - cx, dy = y[sq] - cy, dz = z[sq] - cz;
                close = close || (((dx * dx + dy * dy) + dz * dz) < md2);
              }
              seen += (uint32_t)__builtin_popcountll(km);
            }
            {
              const bool mine = lane < c && my_ok;
              const unsigned long long km = __ballot(mine);
              const uint32_t before = seen + (uint32_t)__builtin_popcountll(km & ((1ull << lane) - 1ull));
              if (mine && before >= lds_points) {
                const float dx = px - cx, dy = py - cy, dz = pz - cz;
                close = close || (((dx * dx + dy * dy) + dz * dz) < md2);
              }
            }
          }
          ok = __ballot(close) == 0ull;
        }
        if (ok) {
          if (lane == 0 && kept < lds_points) { ax[kept] = cx; ay[kept] = cy; az[kept] = cz; }
          kept++;
          wave_sync_lds();
        }
        if (lane == c) my_ok = ok ? 1u : 0u;
      }
      if (in) keep[j] = my_ok;
      if (cnt < 64) break;
    }
  }
}

// MH_KEEP_LDS=<n>: a smaller LDS list (the parity tests drive k_keep_seq's overflow path with it)
inline uint32_t keep_lds_points() {
  const char* e = getenv("MH_KEEP_LDS");
  const long x = e ? atol(e) : (long)kKeepLds;
  return (uint32_t)std::min<long>(std::max<long>(x, 1), (long)kKeepLds);
}

__global__ void k_add_ndt_slots(const uint32_t* __restrict__ head, uint32_t n, uint32_t* __restrict__ keep_plus) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && head[i]) keep_plus[i] += 2u;  // two NDT records in front of the voxel's points
}

// mola::NDT [U] voxel statistics (SURVEY 8a row a13, App.B U10): mean, covariance 1/(n-1), cyclic Jacobi (fp64, the
// same operation sequence as the CPU restatement), plane iff lambda_min/lambda_max < ratio, normal = eigenvector of
// lambda_min with its largest component positive.  Records: pts[first-2] = {centroid, plane flag}, pts[first-1] = {normal, 0}.
__global__ void k_ndt_stats(float4* __restrict__ pts, const uint32_t* __restrict__ vox_first,
                            const uint32_t* __restrict__ vox_count, const uint32_t* __restrict__ n_vox_dev, float max_ratio,
                            uint32_t min_pts, uint32_t* __restrict__ n_planes) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= *n_vox_dev) return;  // (the grid covers the host's upper bound of the voxel count)
  const uint32_t first = vox_first[v], cnt = vox_count[v];
  float4 rc = make_float4(0.f, 0.f, 0.f, 0.f), rn = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cnt >= min_pts) {
    double mu[3] = {0.0, 0.0, 0.0};
    for (uint32_t j = 0; j < cnt; j++) {
      const float4 p = pts[first + j];
      mu[0] += (double)p.x; mu[1] += (double)p.y; mu[2] += (double)p.z;
    }
    mu[0] /= (double)cnt; mu[1] /= (double)cnt; mu[2] /= (double)cnt;
    double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
    for (uint32_t j = 0; j < cnt; j++) {
      const float4 p = pts[first + j];
      const double d0 = (double)p.x - mu[0], d1 = (double)p.y - mu[1], d2 = (double)p.z - mu[2];
      c00 += d0 * d0; c01 += d0 * d1; c02 += d0 * d2; c11 += d1 * d1; c12 += d1 * d2; c22 += d2 * d2;
    }
    const double inv = (double)(cnt - 1);
    double a[3][3] = {{c00 / inv, c01 / inv, c02 / inv}, {c01 / inv, c11 / inv, c12 / inv}, {c02 / inv, c12 / inv, c22 / inv}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 12; sweep++) {
#pragma unroll
      for (int pq = 0; pq < 3; pq++) {
        const int p = (pq == 2) ? 1 : 0, q = (pq == 0) ? 1 : 2, r = 3 - p - q;
        const double apq = a[p][q];
        if (apq != 0.0) {
          const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
          const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
          const double app = a[p][p] - t * apq, aqq = a[q][q] + t * apq;
          const double arp = c * a[r][p] - sn * a[r][q], arq = sn * a[r][p] + c * a[r][q];
          a[p][p] = app; a[q][q] = aqq; a[p][q] = 0.0; a[q][p] = 0.0;
          a[r][p] = arp; a[p][r] = arp; a[r][q] = arq; a[q][r] = arq;
#pragma unroll
          for (int i = 0; i < 3; i++) {
            const double vip = c * V[i][p] - sn * V[i][q], viq = sn * V[i][p] + c * V[i][q];
            V[i][p] = vip; V[i][q] = viq;
          }
        }
      }
    }
    // smallest / largest eigenvalue and the eigenvector of the smallest (stable selection: first minimum)
    const double w0 = a[0][0], w1 = a[1][1], w2 = a[2][2];
    int imin = 0;
    double wmin = w0, wmax = w0;
    if (w1 < wmin) { wmin = w1; imin = 1; }
    if (w2 < wmin) { wmin = w2; imin = 2; }
    if (w1 > wmax) wmax = w1;
    if (w2 > wmax) wmax = w2;
    rc = make_float4((float)mu[0], (float)mu[1], (float)mu[2], 0.f);
    if (wmax > 0.0 && (wmin / wmax) < (double)max_ratio) {
      double nv[3] = {imin == 0 ? V[0][0] : (imin == 1 ? V[0][1] : V[0][2]), imin == 0 ? V[1][0] : (imin == 1 ? V[1][1] : V[1][2]),
                      imin == 0 ? V[2][0] : (imin == 1 ? V[2][1] : V[2][2])};
      const double len = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
      int big = 0;
      if (fabs(nv[1]) > fabs(nv[big])) big = 1;
      if (fabs(nv[2]) > fabs(nv[big])) big = 2;
      const double sgn = ((big == 0 ? nv[0] : (big == 1 ? nv[1] : nv[2])) < 0.0 ? -1.0 : 1.0) / len;
      rn = make_float4((float)(nv[0] * sgn), (float)(nv[1] * sgn), (float)(nv[2] * sgn), 0.f);
      rc.w = 1.f;
      atomicAdd(n_planes, 1u);
    }
  }
  pts[first - 2] = rc;
  pts[first - 1] = rn;
}

// Sub-voxel index for the quad matcher (round-4 experiment): a voxel's records re-ordered, stably, by quadrant
// q = 2 * (x >= mid_x) + (y >= mid_y) of the voxel (mid = (index + 0.5) * voxel size in fp32, the formula the search repeats),
// w = the record's index in `pts`; the quadrants' boundaries packed into one dword at the voxel's hash slot.  Voxels with
// more than 31 records keep their order and get no boundaries (bit 31 clear).
__global__ void k_build_qidx(const float4* __restrict__ pts, const unsigned long long* __restrict__ vox_keys,
                             const uint32_t* __restrict__ vox_first, const uint32_t* __restrict__ vox_count,
                             const uint32_t* __restrict__ n_vox_dev, MapSlot* __restrict__ slots, uint32_t mask, float vs,
                             uint32_t no_index, float4* __restrict__ pts_q) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= *n_vox_dev) return;
  const unsigned long long key = vox_keys[v];
  const uint32_t first = vox_first[v], cnt = vox_count[v];
  // (k_table_insert of this build put it there; should the table have been emptied under this launch -- a later insert's
  //  prologue on another stream -- the key is absent: give up on the voxel instead of probing for ever)
  uint32_t h = hash_key(key) & mask;
  uint32_t probes = 0;
  while (slots[h].key != key) {
    if (slots[h].key == kEmptyKey || ++probes > mask) {
      for (uint32_t j = 0; j < cnt; j++) {  // records in scan order, no boundaries: the voxel is scanned whole
        const float4 p = pts[first + j];
        pts_q[first + j] = make_float4(p.x, p.y, p.z, __uint_as_float(first + j));
      }
      return;
    }
    h = (h + 1) & mask;
  }
  if (cnt > 31u || no_index) {
    for (uint32_t j = 0; j < cnt; j++) {
      const float4 p = pts[first + j];
      pts_q[first + j] = make_float4(p.x, p.y, p.z, __uint_as_float(first + j));
    }
    return;  // (the slot keeps its plain count: scanned whole)
  }
  int kx, ky, kz;
  unpack_key(key, kx, ky, kz);
  const float midx = ((float)kx + 0.5f) * vs, midy = ((float)ky + 0.5f) * vs;
  uint32_t n0 = 0, n1 = 0, n2 = 0;
  for (uint32_t j = 0; j < cnt; j++) {
    const float4 p = pts[first + j];
    const uint32_t q = (p.x >= midx ? 2u : 0u) + (p.y >= midy ? 1u : 0u);
    n0 += q == 0u; n1 += q == 1u; n2 += q == 2u;
  }
  uint32_t o0 = 0, o1 = n0, o2 = n0 + n1, o3 = n0 + n1 + n2;
  slots[h].count = 0x80000000u | (o3 << 18) | (o2 << 13) | (o1 << 8) | cnt;  // (slot_count() in mh_nn_device.h reads it back)
  for (uint32_t j = 0; j < cnt; j++) {
    const float4 p = pts[first + j];
    const uint32_t q = (p.x >= midx ? 2u : 0u) + (p.y >= midy ? 1u : 0u);
    const uint32_t o = q == 0u ? o0 : (q == 1u ? o1 : (q == 2u ? o2 : o3));
    o0 += q == 0u; o1 += q == 1u; o2 += q == 2u; o3 += q == 3u;
    pts_q[first + o] = make_float4(p.x, p.y, p.z, __uint_as_float(first + j));
  }
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
                          uint32_t* __restrict__ counters /* [9], [10]: voxel and record totals, written here */, float4* __restrict__ pts,
                          unsigned long long* __restrict__ vox_keys, uint32_t* __restrict__ vox_first,
                          uint32_t* __restrict__ vox_count, uint32_t* __restrict__ bbox /*6 ordered uints*/,
                          uint32_t ndt, const uint32_t* __restrict__ src_ids /*null: identity*/) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {  // voxel and record totals (the last elements of the two scans) next to the other counters: they stay on
    counters[9] = vid1[n - 1];  // the device for k_ndt_stats / k_table_insert and travel with the one read-back
    counters[10] = outpos[n - 1] + keep[n - 1];
  }
  float px = 0, py = 0, pz = 0;
  bool kept = false;
  if (i < n && (keep[i] & 1u)) {  // bit 0 = point kept (head elements of NDT maps carry +2 for their two records)
    kept = true;
    const uint32_t src = idx_s[i];
    px = x[src]; py = y[src]; pz = z[src];
    const uint32_t pos = outpos[i] + ((ndt && head[i]) ? 2u : 0u);
    pts[pos] = make_float4(px, py, pz, __uint_as_float(src_ids ? src_ids[src] : src));
    if (head[i]) {
      // stored points of this voxel = records between this head and the next one, minus the two NDT records
      const uint32_t v = vid1[i] - 1;
      const uint32_t n_vox = vid1[n - 1];
      const uint32_t total_records = outpos[n - 1] + keep[n - 1];
      const uint32_t next = (v + 1 < n_vox) ? outpos[vstart[v + 1]] : total_records;
      vox_keys[v] = ks[i];
      vox_first[v] = pos;
      vox_count[v] = next - outpos[i] - (ndt ? 2u : 0u);
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
  // one set of atomics per workgroup (one per wave serialised ~4 k waves on six addresses: most of this kernel's time)
  __shared__ uint32_t sh[4][6];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
    for (int a = 0; a < 3; a++) {
      sh[w][a] = mn[a];
      sh[w][3 + a] = mx[a];
    }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = threadIdx.x;
    uint32_t v = sh[0][a];
    for (int q = 1; q < 4; q++) v = a < 3 ? min(v, sh[q][a]) : max(v, sh[q][a]);
    // (only where it moves the box: the value read may be stale, which costs a needless atomic and never a needed one)
    const uint32_t cur = __hip_atomic_load(&bbox[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a < 3) { if (v < cur) atomicMin(&bbox[a], v); }
    else if (v > cur) atomicMax(&bbox[a], v);
  }
}

__global__ void k_table_insert(const unsigned long long* __restrict__ vox_keys, const uint32_t* __restrict__ vox_first,
                               const uint32_t* __restrict__ vox_count, const uint32_t* __restrict__ n_vox_dev,
                               MapSlot* __restrict__ slots, uint32_t mask) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= *n_vox_dev) return;  // (the grid covers the host's upper bound of the voxel count)
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

// mh_map_insert: the stored points, voxel by voxel (ascending key, in-voxel insertion order), back into SoA arrays
__global__ void k_gather_stored(const float4* __restrict__ pts, const uint32_t* __restrict__ vox_first,
                                const uint32_t* __restrict__ vox_count, uint32_t n_vox, uint32_t ndt,
                                float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz,
                                uint32_t* __restrict__ osrc) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n_vox) return;
  const uint32_t first = vox_first[v], cnt = vox_count[v];
  const uint32_t o = ndt ? first - 2u * (v + 1u) : first;  // position in the point-only numbering
  for (uint32_t j = 0; j < cnt; j++) {
    const float4 p = pts[first + j];
    ox[o + j] = p.x;
    oy[o + j] = p.y;
    oz[o + j] = p.z;
    osrc[o + j] = __float_as_uint(p.w);
  }
}

struct Pose12 { double m[12]; };

// FilterMerge with input_layer_in_local_coordinates: p_map = (float)(R*p + t) (CPose3D::composePoint [U])
__global__ void k_compose_new(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                              uint32_t n, Pose12 T, uint32_t src0, float* __restrict__ ox, float* __restrict__ oy,
                              float* __restrict__ oz, uint32_t* __restrict__ osrc) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double px = x[i], py = y[i], pz = z[i];
  ox[i] = (float)(((T.m[0] * px + T.m[1] * py) + T.m[2] * pz) + T.m[3]);
  oy[i] = (float)(((T.m[4] * px + T.m[5] * py) + T.m[6] * pz) + T.m[7]);
  oz[i] = (float)(((T.m[8] * px + T.m[9] * py) + T.m[10] * pz) + T.m[11]);
  osrc[i] = src0 + i;
}

inline uint32_t nblk(size_t n, uint32_t b) { return (uint32_t)((n + b - 1) / b); }

}  // namespace

extern "C" {

mh_status mh_map_create(mh_ctx* ctx, const mh_map_params* params, mh_map** out) {
  MH_REQUIRE(ctx && params && out, "null argument");
  *out = nullptr;
  MH_REQUIRE(params->voxel_size > 0.f && isfinite(params->voxel_size), "voxel_size must be > 0");
  MH_REQUIRE(params->index_mode == MH_INDEX_FLOOR || params->index_mode == MH_INDEX_TRUNC, "bad index_mode");
  MH_REQUIRE(params->far_voxel_metric <= MH_FAR_L2, "bad far_voxel_metric");
  MH_REQUIRE(params->min_distance_between_points >= 0.f && params->ndt_max_eigen_ratio >= 0.f, "negative NDT parameter");
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
  (void)mh::wait_stream(m->ctx->stream);
  if (m->ev_counts) (void)hipEventDestroy(m->ev_counts);
  if (m->ev_qidx) (void)hipEventDestroy(m->ev_qidx);
  if (m->h_counts) (void)hipHostFree(m->h_counts);
  for (mh::DevBuf* b : {&m->build_a, &m->build_b, &m->build_c, &m->build_d, &m->build_e, &m->sort_tmp}) b->release();
  m->slots.release();
  m->pts.release();
  m->vox_keys.release();
  m->vox_first.release();
  m->vox_count.release();
  m->merge.release();
  m->pts_q.release();
  delete m;
  return MH_OK;
}

// mh_map_insert, merge path: the two kernels above and k_keys in ONE launch.  The first workgroups walk the stored voxels
// (a stored point's key is its voxel's key: the same function of the same coordinates gave it when it was inserted), the
// others compose the new layer with the pose and key it the way k_keys does; everybody writes idx[i] = i.
__global__ __launch_bounds__(256) void k_collect(const float4* __restrict__ pts, const unsigned long long* __restrict__ vox_keys,
                                                 const uint32_t* __restrict__ vox_first, const uint32_t* __restrict__ vox_count,
                                                 uint32_t n_vox, uint32_t ndt, uint32_t stored_blocks, const float* __restrict__ x,
                                                 const float* __restrict__ y, const float* __restrict__ z, uint32_t n_new, Pose12 T,
                                                 uint32_t src0, uint32_t n_old, float inv_vs, uint32_t trunc,
                                                 float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz,
                                                 uint32_t* __restrict__ osrc, unsigned long long* __restrict__ keys,
                                                 uint32_t* __restrict__ idx, uint32_t* __restrict__ flags) {
  if (blockIdx.x < stored_blocks) {
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    if (v >= n_vox) return;
    const uint32_t first = vox_first[v], cnt = vox_count[v];
    const unsigned long long key = vox_keys[v];
    const uint32_t o = ndt ? first - 2u * (v + 1u) : first;  // position in the point-only numbering
    for (uint32_t j = 0; j < cnt; j++) {
      const float4 p = pts[first + j];
      ox[o + j] = p.x;
      oy[o + j] = p.y;
      oz[o + j] = p.z;
      osrc[o + j] = __float_as_uint(p.w);
      keys[o + j] = key;
      idx[o + j] = o + j;
    }
    return;
  }
  const uint32_t i = (blockIdx.x - stored_blocks) * 256 + threadIdx.x;
  if (i >= n_new) return;
  const double lx = x[i], ly = y[i], lz = z[i];
  const float px = (float)(((T.m[0] * lx + T.m[1] * ly) + T.m[2] * lz) + T.m[3]);
  const float py = (float)(((T.m[4] * lx + T.m[5] * ly) + T.m[6] * lz) + T.m[7]);
  const float pz = (float)(((T.m[8] * lx + T.m[9] * ly) + T.m[10] * lz) + T.m[11]);
  const uint32_t o = n_old + i;
  ox[o] = px;
  oy[o] = py;
  oz[o] = pz;
  osrc[o] = src0 + i;
  unsigned long long k = kEmptyKey;
  if (isfinite(px) && isfinite(py) && isfinite(pz)) {
    const float sx = px * inv_vs, sy = py * inv_vs, sz = pz * inv_vs;
    if (fabsf(sx) < 1.0e6f && fabsf(sy) < 1.0e6f && fabsf(sz) < 1.0e6f)
      k = pack_key(voxel_index(px, inv_vs, trunc), voxel_index(py, inv_vs, trunc), voxel_index(pz, inv_vs, trunc));
    else
      atomicOr(&flags[0], 1u);  // voxel index does not fit 21 bits
  }
  keys[o] = k;
  idx[o] = o;
}

mh_status mh_map_build(mh_map* m, const float* x, const float* y, const float* z, size_t n, int32_t mem) {
  MH_REQUIRE(m, "null map");
  MH_REQUIRE(mem == MH_MEM_HOST || mem == MH_MEM_DEVICE, "bad mem space");
  MH_REQUIRE(n == 0 || (x && y && z), "null point arrays");
  MH_REQUIRE(n < 0x7FFFFFF0ull, "too many points");
  mh_ctx* ctx = m->ctx;
  MH_TRY(set_device(ctx));
  hipStream_t s = ctx->stream;
  MH_HIP(mh::wait_stream(s));  // a rebuild invalidates everything queued against the old content
  (void)map_resolve(m);  // (a pending verdict about the OLD content is moot now)
  const float *dx = x, *dy = y, *dz = z;
  if (n > 0 && mem == MH_MEM_HOST) {
    const size_t stride = ((n * sizeof(float) + 255) / 256) * 256;
    MH_TRY(ctx->staging.reserve(3 * stride));
    MH_TRY(stage_in(ctx, ctx->staging, 0, x, n * sizeof(float), mem));
    MH_TRY(stage_in(ctx, ctx->staging, stride, y, n * sizeof(float), mem));
    MH_TRY(stage_in(ctx, ctx->staging, 2 * stride, z, n * sizeof(float), mem));
    dx = (const float*)ctx->staging.as<char>();
    dy = (const float*)(ctx->staging.as<char>() + stride);
    dz = (const float*)(ctx->staging.as<char>() + 2 * stride);
  }
  MH_TRY(map_build_device(m, s, dx, dy, dz, nullptr, n, nullptr, 0));
  m->n_offered = n;
  return map_resolve(m);  // a build reports its own verdict (out-of-range points) and leaves nothing in flight
}

mh_status mh_map_insert(mh_map* m, const mh_scan* scan, const double T[12], float remove_voxels_farther_than) {
  MH_REQUIRE(m && scan && T, "null argument");
  MH_REQUIRE(m->ctx == scan->ctx, "map and scan belong to different contexts");
  MH_REQUIRE(remove_voxels_farther_than >= 0.f, "negative remove_voxels_farther_than");
  for (int i = 0; i < 12; i++) MH_REQUIRE(isfinite(T[i]), "non-finite pose");
  mh_ctx* ctx = m->ctx;
  MH_TRY(set_device(ctx));
  // the previous update's counts (long complete by now: an align has run in between).  Its out-of-range verdict, if any, is
  // reported by THIS call -- after this call's own insertion has been performed: a valid key-frame is never dropped because
  // the one before it held a wild point.
  MH_TRY(map_resolve_counts(m));
  const bool prev_out_of_range = m->deferred_error != MH_OK;
  m->deferred_error = MH_OK;
  // The update is asynchronous (no host synchronisation: counts and verdict are read back lazily) and runs on the context's
  // stream.  (Round 3 also ran it on a stream of the map's own, MH_MAP_SIDE_STREAM=1, waited for by the next use of the map
  // only: the extra event traffic cost the caller more -- 0.139 ms per key-frame against 0.101 -- than the overlap with the
  // next scan's de-skew returned, 967 against 990 scans/s; removed in round 4.)
  hipStream_t s = ctx->stream;
  const size_t n_old = m->n_points, n_new = scan->n, total = n_old + n_new;
  MH_REQUIRE(total < 0x7FFFFFF0ull && m->n_offered + n_new < 0xFFFFFFF0ull, "too many points");
  const size_t stride = ((total * sizeof(float) + 255) / 256) * 256;
  MH_TRY(m->merge.reserve(4 * stride ? 4 * stride : 256));
  char* base = m->merge.as<char>();
  float *mx = (float*)base, *my = (float*)(base + stride), *mz = (float*)(base + 2 * stride);
  uint32_t* msrc = (uint32_t*)(base + 3 * stride);
  const uint32_t ndt = m->params.ndt_max_eigen_ratio > 0.f ? 1u : 0u;
  // merge path (stored points present, new points to add): gather + compose + keys in one launch, behind the rebuild's
  // own first launch (counters and slot table) -- map_build_device is told that both have been done
  // (up to ~0.4 M stored points: beyond, the voxel-by-voxel walk that also writes keys loses to the streaming k_keys --
  // a 10 k-point key-frame into 0.1 / 0.25 / 0.5 / 1 M points completes after 0.117 / 0.139 / 0.180 / 0.272 ms fused and
  // 0.132 / 0.156 / 0.181 / 0.242 ms with the three launches)
  const bool fused_collect = m->n_voxels && n_new && n_old <= 400000 && getenv("MH_MAP_FULL_SORT") == nullptr &&
                             getenv("MH_MAP_NO_COLLECT") == nullptr;
  if (fused_collect) {
    uint32_t* counters = nullptr;
    unsigned long long* keys = nullptr;
    uint32_t* idx = nullptr;
    MH_TRY(map_build_prologue(m, s, total, n_old, &counters, &keys, &idx));
    Pose12 P;
    for (int i = 0; i < 12; i++) P.m[i] = T[i];
    const uint32_t stored_blocks = nblk(m->n_voxels, 256);
    hipLaunchKernelGGL(k_collect, dim3(stored_blocks + nblk(n_new, 256)), dim3(256), 0, s, m->pts.as<float4>(),
                       m->vox_keys.as<unsigned long long>(), m->vox_first.as<uint32_t>(), m->vox_count.as<uint32_t>(),
                       (uint32_t)m->n_voxels, ndt, stored_blocks, scan->x, scan->y, scan->z, (uint32_t)n_new, P,
                       (uint32_t)m->n_offered, (uint32_t)n_old, m->inv_vs, (uint32_t)(m->params.index_mode == MH_INDEX_TRUNC), mx, my,
                       mz, msrc, keys, idx, counters);
  } else {
    if (m->n_voxels)
      hipLaunchKernelGGL(k_gather_stored, dim3(nblk(m->n_voxels, 128)), dim3(128), 0, s, m->pts.as<float4>(),
                         m->vox_first.as<uint32_t>(), m->vox_count.as<uint32_t>(), (uint32_t)m->n_voxels, ndt, mx, my, mz,
                         msrc);
    if (n_new) {
      Pose12 P;
      for (int i = 0; i < 12; i++) P.m[i] = T[i];
      hipLaunchKernelGGL(k_compose_new, dim3(nblk(n_new, 256)), dim3(256), 0, s, scan->x, scan->y, scan->z, (uint32_t)n_new,
                         P, (uint32_t)m->n_offered, mx + n_old, my + n_old, mz + n_old, msrc + n_old);
    }
  }
  MH_HIP(hipGetLastError());
  int evict[4] = {0, 0, 0, -1};
  if (remove_voxels_farther_than > 0.f) {
    const bool trunc = m->params.index_mode == MH_INDEX_TRUNC;
    for (int a = 0; a < 3; a++) {
      const float sc = (float)T[4 * a + 3] * m->inv_vs;
      MH_REQUIRE(fabsf(sc) < 1.0e6f, "insertion pose outside the key range");
      evict[a] = trunc ? (int)sc : (int)floorf(sc);
    }
    evict[3] = (int)ceilf(remove_voxels_farther_than * m->inv_vs);
  }
  MH_TRY(map_build_device(m, s, mx, my, mz, msrc, total, evict, n_old, fused_collect));
  m->n_offered += n_new;
  if (prev_out_of_range)  // a status of its own, not a failure (a caller must be able to tell "inserted" from "not inserted")
    return fail(MH_WARN_PREVIOUS_OUT_OF_RANGE, "the PREVIOUS update of this map held points whose voxel index exceeds the +-2^20 range of the "
                                               "packed key (|coord|/voxel_size must be < 1e6); they were left out.  This call's insertion HAS been performed");
  return MH_OK;
}

}  // extern "C"

namespace mh {

// counters: [0] range flag  [1] n_valid  [2..7] bbox (ordered uints)  [8] planes  [9] voxels  [10] records
// ... and the empty slot table of the rebuild, in the same launch
__global__ __launch_bounds__(256) void k_init_build(uint32_t* __restrict__ c, uint4* __restrict__ slots, uint32_t n_slots) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid < 16) c[tid] = (tid >= 2 && tid <= 4) ? 0xFFFFFFFFu : 0u;
  const uint4 e = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
  for (uint32_t i = tid; i < n_slots; i += gridDim.x * blockDim.x) slots[i] = e;
}

// Counts, bounding box and the out-of-range flag of the last (re)build, once its read-back has arrived.  The verdict is
// only RECORDED here (m->deferred_error); who reports it is the caller's business: mh_map_build returns its own,
// mh_map_insert returns the PREVIOUS update's after having performed the new one, the accessors never fail for it
// (mh_map_info::deferred_status shows it).
mh_status map_resolve_counts(const mh_map* m) {
  if (m->counts_pending) {
    MH_HIP(mh::wait_event(m->ev_counts));
    m->counts_pending = false;
    m->build_in_flight = false;
    const uint32_t* h = m->h_counts;
    const bool ndt = m->params.ndt_max_eigen_ratio > 0.f;
    m->n_voxels = h[9];
    m->n_records = h[10];
    m->n_points = h[10] - (ndt ? 2u * h[9] : 0u);
    m->n_planes = m->n_points ? h[8] : 0;
    for (int a = 0; a < 3; a++) {
      m->bbox_min[a] = m->n_points ? ord2f(h[2 + a]) : 0.f;
      m->bbox_max[a] = m->n_points ? ord2f(h[5 + a]) : 0.f;
    }
    if (h[0] & 1u) m->deferred_error = MH_ERR_OUT_OF_RANGE;
  }
  return MH_OK;
}

// ... and hand the recorded verdict out (once)
mh_status map_take_verdict(const mh_map* m, const char* whose) {
  if (m->deferred_error == MH_OK) return MH_OK;
  const mh_status e = m->deferred_error;
  m->deferred_error = MH_OK;
  char msg[320];
  snprintf(msg, sizeof msg, "%s: a point's voxel index exceeds the +-2^20 range of the packed key (|coord|/voxel_size must be < 1e6); "
                            "the offending points were left out of the map", whose);
  return fail(e, msg);
}

mh_status map_resolve(const mh_map* m) {
  MH_TRY(map_resolve_counts(m));
  return map_take_verdict(m, "map update");
}

// The sub-voxel index of the quad matcher (MapView::pts_q / qidx, k_build_qidx), built on first use after a (re)build, on the
// stream that is about to search (the caller has ordered that stream behind the map's build: map_ready_on).  Other streams that
// come later wait for the event; MH_NO_QIDX=1 builds it without boundaries (every voxel scanned whole: the A/B baseline).
mh_status map_ensure_qidx(const mh_map* mc, hipStream_t s) {
  mh_map* m = const_cast<mh_map*>(mc);
  std::lock_guard<std::mutex> lk(m->qidx_mtx);
  if (m->qidx_valid) {
    if (m->qidx_pending && s != m->qidx_stream) {
      if (hipEventQuery(m->ev_qidx) == hipSuccess) m->qidx_pending = false;
      else MH_HIP(hipStreamWaitEvent(s, m->ev_qidx, 0));
    }
    return MH_OK;
  }
  if (!m->ev_qidx) MH_HIP(hipEventCreateWithFlags(&m->ev_qidx, hipEventDisableTiming));
  const size_t tsize = m->table_size ? m->table_size : 1;
  MH_TRY(m->pts_q.reserve(m->pts.bytes ? m->pts.bytes : sizeof(float4)));
  if (m->n_voxels && m->d_counters) {
    const uint32_t no_index = (m->params.index_mode == MH_INDEX_TRUNC || getenv("MH_NO_QIDX") != nullptr) ? 1u : 0u;
    hipLaunchKernelGGL(k_build_qidx, dim3(nblk(m->n_voxels, 128)), dim3(128), 0, s, m->pts.as<float4>(),
                       m->vox_keys.as<unsigned long long>(), m->vox_first.as<uint32_t>(), m->vox_count.as<uint32_t>(),
                       m->d_counters + 9, m->slots.as<MapSlot>(), (uint32_t)(tsize - 1), 1.0f / m->inv_vs, no_index,
                       m->pts_q.as<float4>());
    MH_HIP(hipGetLastError());
  }
  MH_HIP(hipEventRecord(m->ev_qidx, s));
  m->qidx_stream = s;
  m->qidx_pending = true;
  m->qidx_valid = true;
  return MH_OK;
}

mh_status map_ready_on(const mh_map* m, hipStream_t s) {
  if (!m->build_in_flight) return MH_OK;
  if (hipEventQuery(m->ev_counts) == hipSuccess) {
    m->build_in_flight = false;
    return MH_OK;
  }
  if (s != m->ctx->stream) MH_HIP(hipStreamWaitEvent(s, m->ev_counts, 0));  // (an alignment on another context's stream)
  return MH_OK;
}

// The new points of a key-frame (a layer of ~10 k points) sorted by voxel key, stably.  rocprim's radix sort takes its
// merge-sort route at this size with 1024 keys per workgroup: one block sort + three or four merge passes + up to two
// copies, six or seven launches of ~5 us each on a path where the launches are the cost.  The same stable merge sort
// with 4096 keys per workgroup needs one block sort and one or two merge passes; larger layers keep the radix sort.
static hipError_t sort_new_points(void* tmp, size_t& tmp_bytes, unsigned long long* keys_in, unsigned long long* keys_out,
                                  uint32_t* idx_in, uint32_t* idx_out, size_t n, hipStream_t s) {
  static const bool radix_only = getenv("MH_MAP_RADIX_SORT_NEW") != nullptr;
  if (n > 65536 || radix_only) return rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, idx_in, idx_out, n, 0, 64, s);
  using cfg = rocprim::merge_sort_config<512, 512, 8>;  // (8192 per workgroup: the block sort alone takes 92 us instead of 22)
  return rocprim::merge_sort<cfg>(tmp, tmp_bytes, keys_in, keys_out, idx_in, idx_out, n, rocprim::less<unsigned long long>(), s);
}

// What a rebuild needs before its first kernel: sizes from bounds the host knows, the scratch and the slot table reserved,
// counters and table initialised (one launch).  mh_map_insert's merge path calls it ahead of its fused collect kernel.
mh_status map_build_prologue(mh_map* m, hipStream_t s, size_t n, size_t n_stored, uint32_t** counters_out,
                             unsigned long long** keys_out, uint32_t** idx_out) {
  if (!m->h_counts) {
    MH_HIP(hipHostMalloc((void**)&m->h_counts, 16 * sizeof(uint32_t), hipHostMallocDefault));
    MH_HIP(hipEventCreateWithFlags(&m->ev_counts, hipEventDisableTiming));
  }
  // what the host can know without waiting for the device: upper bounds (every offered point kept, every new point a voxel)
  const size_t n_vox_ub = n_stored ? std::min<size_t>(n, m->n_voxels + (n - n_stored)) : n;
  uint64_t tsize = 64;  // hash table: power of two, load factor <= 0.5
  while (tsize < 2ull * n_vox_ub) tsize <<= 1;
  MH_TRY(m->build_e.reserve((n ? n : 1) * sizeof(uint32_t) + 64));  // vstart | counters(16)
  uint32_t* counters = m->build_e.as<uint32_t>() + (n ? n : 1);
  MH_TRY(m->slots.reserve(tsize * sizeof(MapSlot)));
  if (n > 0) {
    const size_t n_new = n - n_stored;
    MH_TRY(m->build_a.reserve((2 * n + n_new) * sizeof(unsigned long long)));  // keys in | keys sorted | new keys sorted
    MH_TRY(m->build_b.reserve((2 * n + n_new) * sizeof(uint32_t)));            // idx in | idx sorted | new idx sorted
  }
  {  // the table is about to be emptied: the sub-voxel index of the old build, and the counters it was built from, are void NOW
     // (not only when the new build has been queued: a failure in between must not leave a "valid" index over an empty table)
    std::lock_guard<std::mutex> lk(m->qidx_mtx);
    m->qidx_valid = false;
    m->qidx_pending = false;
    m->d_counters = nullptr;
  }
  {
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((tsize + 255) / 256, 2048);
    hipLaunchKernelGGL(k_init_build, dim3(blocks), dim3(256), 0, s, counters, reinterpret_cast<uint4*>(m->slots.p), (uint32_t)tsize);
  }
  if (counters_out) *counters_out = counters;
  if (keys_out) *keys_out = m->build_a.as<unsigned long long>();
  if (idx_out) *idx_out = m->build_b.as<uint32_t>();
  return MH_OK;
}

mh_status map_build_device(mh_map* m, hipStream_t s, const float* dx, const float* dy, const float* dz, const uint32_t* dsrc,
                           size_t n, const int* evict, size_t n_stored, bool collected) {
  const uint32_t ndt = m->params.ndt_max_eigen_ratio > 0.f ? 1u : 0u;
  const size_t n_vox_ub = n_stored ? std::min<size_t>(n, m->n_voxels + (n - n_stored)) : n;
  const size_t n_rec_ub = n + (ndt ? 2 * n_vox_ub : 0);
  uint64_t tsize = 64;
  while (tsize < 2ull * n_vox_ub) tsize <<= 1;
  // The prologue empties the slot table before the remaining buffers are reserved (their sizes are only bounds).  Should one of
  // those reserves fail (out of memory), the map must not be left describing its OLD content over an EMPTY table -- later
  // alignments would silently find no pairings in a map that claims a million points: it reads as empty instead.
  struct EmptyOnError {
    mh_map* m;
    bool ok = false;
    ~EmptyOnError() {
      if (ok) return;
      m->n_points = m->n_voxels = m->n_records = m->n_planes = 0;
      m->counts_pending = false;
      m->build_in_flight = false;
    }
  } guard{m};
  // (`collected`: mh_map_insert has run the prologue and its fused kernel has written the points, keys and indices)
  if (!collected) MH_TRY(map_build_prologue(m, s, n, n_stored, nullptr, nullptr, nullptr));
  uint32_t* vstart = m->build_e.as<uint32_t>();
  uint32_t* counters = vstart + (n ? n : 1);
  if (n > 0) {
    const uint32_t N = (uint32_t)n;
    // scratch carve-up
    // mh_map_insert hands over [stored points, voxel by voxel in key order | new points]: only the new ones need sorting,
    // one merge puts them in place (stored first among equal keys = insertion order).  MH_MAP_FULL_SORT=1: sort everything.
    const size_t n_new = n - n_stored;
    const bool merge_path = n_stored > 0 && getenv("MH_MAP_FULL_SORT") == nullptr;
    MH_TRY(m->build_a.reserve((2 * n + n_new) * sizeof(unsigned long long)));  // keys in | keys sorted | new keys sorted
    MH_TRY(m->build_b.reserve((2 * n + n_new) * sizeof(uint32_t)));            // idx in | idx sorted | new idx sorted
    MH_TRY(m->build_c.reserve(2 * n * sizeof(uint32_t)));            // head | vid1
    MH_TRY(m->build_d.reserve(2 * n * sizeof(uint32_t)));            // keep | outpos
    unsigned long long* keys = m->build_a.as<unsigned long long>();
    unsigned long long* keys_s = keys + n;
    uint32_t* idx = m->build_b.as<uint32_t>();
    uint32_t* idx_s = idx + n;
    uint32_t* head = m->build_c.as<uint32_t>();
    uint32_t* vid1 = head + n;
    uint32_t* keep = m->build_d.as<uint32_t>();
    uint32_t* outpos = keep + n;

    const uint32_t B = 256;
    const int4 ev = evict ? make_int4(evict[0], evict[1], evict[2], evict[3]) : make_int4(0, 0, 0, -1);
    const int4 ev_keys = merge_path ? make_int4(0, 0, 0, -1) : ev;  // (merge path: eviction after the merge, inside k_heads)
    if (!collected)
      hipLaunchKernelGGL(k_keys, dim3(nblk(n, B)), dim3(B), 0, s, dx, dy, dz, N, m->inv_vs,
                         (uint32_t)(m->params.index_mode == MH_INDEX_TRUNC), ev_keys, m->params.far_voxel_metric, keys, idx, counters);
    unsigned long long* keys_new = keys + 2 * n;
    uint32_t* idx_new = idx + 2 * n;
    size_t tmp = 0;
    if (merge_path) {
      size_t t2 = 0;
      if (n_new) MH_HIP(sort_new_points(nullptr, tmp, keys + n_stored, keys_new, idx + n_stored, idx_new, n_new, s));
      MH_HIP(rocprim::merge(nullptr, t2, keys, keys_new, keys_s, idx, idx_new, idx_s, n_stored, n_new,
                            rocprim::less<unsigned long long>(), s));
      if (t2 > tmp) tmp = t2;
    } else {
      MH_HIP(rocprim::radix_sort_pairs(nullptr, tmp, keys, keys_s, idx, idx_s, N, 0, 64, s));
    }
    {
      size_t t2 = 0;
      MH_HIP(rocprim::inclusive_scan(nullptr, t2, head, vid1, N, rocprim::plus<uint32_t>(), s));
      if (t2 > tmp) tmp = t2;
      MH_HIP(rocprim::exclusive_scan(nullptr, t2, keep, outpos, 0u, N, rocprim::plus<uint32_t>(), s));
      if (t2 > tmp) tmp = t2;
    }
    MH_TRY(m->sort_tmp.reserve(tmp));
    size_t tb = m->sort_tmp.bytes;
    if (merge_path) {
      if (n_new) MH_HIP(sort_new_points(m->sort_tmp.p, tb, keys + n_stored, keys_new, idx + n_stored, idx_new, n_new, s));
      tb = m->sort_tmp.bytes;
      MH_HIP(rocprim::merge(m->sort_tmp.p, tb, keys, keys_new, keys_s, idx, idx_new, idx_s, n_stored, n_new,
                            rocprim::less<unsigned long long>(), s));
    } else {
      MH_HIP(rocprim::radix_sort_pairs(m->sort_tmp.p, tb, keys, keys_s, idx, idx_s, N, 0, 64, s));
    }
    hipLaunchKernelGGL(k_heads, dim3(nblk(n, B)), dim3(B), 0, s, keys_s, N, head, counters,
                       merge_path ? ev : make_int4(0, 0, 0, -1), m->params.far_voxel_metric);  // (merge path: eviction on the sorted keys)
    tb = m->sort_tmp.bytes;
    MH_HIP(rocprim::inclusive_scan(m->sort_tmp.p, tb, head, vid1, N, rocprim::plus<uint32_t>(), s));
    hipLaunchKernelGGL(k_vstart, dim3(nblk(n, B)), dim3(B), 0, s, head, vid1, N, vstart);
    if (m->params.min_distance_between_points > 0.f)
      hipLaunchKernelGGL(k_keep_seq, dim3(nblk(n, kKeepBlock)), dim3(kKeepBlock), 0, s, dx, dy, dz, keys_s, idx_s, head, vid1,
                         vstart, N, m->params.max_points_per_voxel, m->params.min_distance_between_points, (uint32_t)n_stored,
                         keep_lds_points(), keep);
    else
      hipLaunchKernelGGL(k_keep, dim3(nblk(n, B)), dim3(B), 0, s, keys_s, vid1, vstart, N, m->params.max_points_per_voxel,
                         keep);
    if (ndt) hipLaunchKernelGGL(k_add_ndt_slots, dim3(nblk(n, B)), dim3(B), 0, s, head, N, keep);
    tb = m->sort_tmp.bytes;
    MH_HIP(rocprim::exclusive_scan(m->sort_tmp.p, tb, keep, outpos, 0u, N, rocprim::plus<uint32_t>(), s));

    MH_TRY(m->pts.reserve(n_rec_ub * sizeof(float4)));
    MH_TRY(m->vox_keys.reserve(n_vox_ub * sizeof(unsigned long long)));
    MH_TRY(m->vox_first.reserve(n_vox_ub * sizeof(uint32_t)));
    MH_TRY(m->vox_count.reserve(n_vox_ub * sizeof(uint32_t)));
    hipLaunchKernelGGL(k_scatter, dim3(nblk(n, B)), dim3(B), 0, s, dx, dy, dz, keys_s, idx_s, head, vid1, vstart, keep,
                       outpos, N, m->params.max_points_per_voxel, counters, m->pts.as<float4>(),
                       m->vox_keys.as<unsigned long long>(), m->vox_first.as<uint32_t>(), m->vox_count.as<uint32_t>(),
                       counters + 2, ndt, dsrc);
    if (ndt)
      hipLaunchKernelGGL(k_ndt_stats, dim3(nblk(n_vox_ub, 128)), dim3(128), 0, s, m->pts.as<float4>(),
                         m->vox_first.as<uint32_t>(), m->vox_count.as<uint32_t>(), counters + 9, m->params.ndt_max_eigen_ratio,
                         m->params.ndt_min_points ? m->params.ndt_min_points : 4u, counters + 8);
    hipLaunchKernelGGL(k_table_insert, dim3(nblk(n_vox_ub, 256)), dim3(256), 0, s, m->vox_keys.as<unsigned long long>(),
                       m->vox_first.as<uint32_t>(), m->vox_count.as<uint32_t>(), counters + 9, m->slots.as<MapSlot>(),
                       (uint32_t)(tsize - 1));
  }
  {  // the quad matcher's sub-voxel index follows the records lazily (map_ensure_qidx: only alignments of large layers need it)
    std::lock_guard<std::mutex> lk(m->qidx_mtx);
    m->qidx_valid = false;
    m->qidx_pending = false;
    m->d_counters = counters;
  }
  MH_HIP(hipGetLastError());
  MH_HIP(hipMemcpyAsync(m->h_counts, counters, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));  // the one read-back, lazy
  MH_HIP(hipEventRecord(m->ev_counts, s));
  m->counts_pending = true;
  m->build_in_flight = true;
  m->table_size = tsize;
  // until map_resolve(): bounds, good enough for whoever only sizes buffers
  m->n_points = n;
  m->n_voxels = n_vox_ub;
  m->n_records = n_rec_ub;
  guard.ok = true;
  return MH_OK;
}

}  // namespace mh

extern "C" {

mh_status mh_map_get_info(const mh_map* m, mh_map_info* info) {
  MH_REQUIRE(m && info, "null argument");
  MH_TRY(set_device(m->ctx));
  MH_TRY(map_resolve_counts(m));  // (never fails for a deferred verdict: see deferred_status)
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
  info->n_planes = m->n_planes;
  info->deferred_status = (uint32_t)m->deferred_error;
  return MH_OK;
}

mh_status mh_map_download(const mh_map* m, float* x, float* y, float* z, uint32_t* src_idx, int32_t* vox_keys_xyz,
                          uint32_t* vox_first, uint32_t* vox_count) {
  MH_REQUIRE(m, "null map");
  mh_ctx* ctx = m->ctx;
  MH_TRY(set_device(ctx));
  MH_HIP(mh::wait_stream(ctx->stream));
  MH_TRY(map_resolve_counts(m));
  if (!m->n_voxels) return MH_OK;
  std::vector<uint32_t> hf(m->n_voxels), hc(m->n_voxels);
  MH_HIP(hipMemcpy(hf.data(), m->vox_first.p, m->n_voxels * 4, hipMemcpyDeviceToHost));
  MH_HIP(hipMemcpy(hc.data(), m->vox_count.p, m->n_voxels * 4, hipMemcpyDeviceToHost));
  if (x || y || z || src_idx) {
    std::vector<float4> h(m->n_records);
    MH_HIP(hipMemcpy(h.data(), m->pts.p, m->n_records * sizeof(float4), hipMemcpyDeviceToHost));
    size_t o = 0;
    for (size_t v = 0; v < m->n_voxels; v++)  // NDT maps interleave two statistics records per voxel: skip them
      for (uint32_t j = 0; j < hc[v]; j++, o++) {
        const float4& p = h[hf[v] + j];
        if (x) x[o] = p.x;
        if (y) y[o] = p.y;
        if (z) z[o] = p.z;
        if (src_idx) memcpy(&src_idx[o], &p.w, 4);
      }
  }
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
  size_t o = 0;
  for (size_t v = 0; v < m->n_voxels; v++) {  // offsets in the point-only numbering of the arrays above
    if (vox_first) vox_first[v] = (uint32_t)o;
    if (vox_count) vox_count[v] = hc[v];
    o += hc[v];
  }
  return MH_OK;
}

mh_status mh_map_download_ndt(const mh_map* m, float* cx, float* cy, float* cz, float* nx, float* ny, float* nz,
                              uint32_t* is_plane) {
  MH_REQUIRE(m, "null map");
  mh_ctx* ctx = m->ctx;
  MH_TRY(set_device(ctx));
  MH_HIP(mh::wait_stream(ctx->stream));
  MH_TRY(map_resolve_counts(m));
  if (!m->n_voxels) return MH_OK;
  const bool ndt = m->params.ndt_max_eigen_ratio > 0.f;
  std::vector<uint32_t> hf(m->n_voxels);
  std::vector<float4> h;
  if (ndt) {
    MH_HIP(hipMemcpy(hf.data(), m->vox_first.p, m->n_voxels * 4, hipMemcpyDeviceToHost));
    h.resize(m->n_records);
    MH_HIP(hipMemcpy(h.data(), m->pts.p, m->n_records * sizeof(float4), hipMemcpyDeviceToHost));
  }
  for (size_t v = 0; v < m->n_voxels; v++) {
    const float4 c = ndt ? h[hf[v] - 2] : make_float4(0, 0, 0, 0), nn = ndt ? h[hf[v] - 1] : make_float4(0, 0, 0, 0);
    if (cx) cx[v] = c.x;
    if (cy) cy[v] = c.y;
    if (cz) cz[v] = c.z;
    if (nx) nx[v] = nn.x;
    if (ny) ny[v] = nn.y;
    if (nz) nz[v] = nn.z;
    if (is_plane) is_plane[v] = c.w != 0.f ? 1u : 0u;
  }
  return MH_OK;
}

}  // extern "C"
