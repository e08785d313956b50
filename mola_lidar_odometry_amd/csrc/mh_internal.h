// mh_internal.h -- private structures shared by the libmolahip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>

#include "../../include/molahip.h"
#include "mh_se3.h"

namespace mh {

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const char* fmt, ...);
mh_status fail(mh_status s, const char* fmt, ...);

#define MH_HIP(expr)                                                                                  \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess)                                                                             \
      return mh::fail(_e == hipErrorOutOfMemory ? MH_ERR_OUT_OF_MEMORY : MH_ERR_HIP, "%s failed: %s (%s:%d)", \
                      #expr, hipGetErrorString(_e), __FILE__, __LINE__);                              \
  } while (0)

#define MH_TRY(expr)               \
  do {                             \
    mh_status _s = (expr);         \
    if (_s != MH_OK) return _s;    \
  } while (0)

#define MH_REQUIRE(cond, msg)                                                     \
  do {                                                                            \
    if (!(cond)) return mh::fail(MH_ERR_INVALID_ARGUMENT, "%s: %s", __func__, msg); \
  } while (0)

// grow-only device buffer
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  // Growing: capacity doubles, and the block that became too small is RETIRED, not freed -- hipFree waits for the whole
  // device (every stream of every sequence of the process: measured 45 us alone, 400 us with eight sequences) and a map
  // that grows by a key-frame per scan outgrew one of its dozen buffers twice per scan with 25 % headroom.  Retired blocks
  // add up to less than the live one (geometric series) and are returned by release(); work still in flight on the old
  // block keeps reading valid memory, so no caller has to drain a stream before growing.
  void* retired[24] = {};
  int n_retired = 0;
  mh_status reserve(size_t need) {
    if (need <= bytes) return MH_OK;
    // head-room: double while the block is small (a map that grows by a key-frame per scan would otherwise outgrow a dozen
    // buffers every few scans), a quarter beyond 64 MB -- doubling a 10 GB map layer on a device that holds several is how a
    // process runs out of memory with most of it unused
    const size_t cap = (need > (size_t(64) << 20) ? need + need / 4 : 2 * need) + 256;
    void* q = nullptr;
    hipError_t e = fault_first_attempts.load() > 0 && fault_first_attempts.fetch_sub(1) > 0 ? hipErrorOutOfMemory : hipMalloc(&q, cap);
    if (e != hipSuccess) {  // under memory pressure: give the retired blocks back and ask for what is needed only
      free_retired();
      e = fault_retries.load() > 0 && fault_retries.fetch_sub(1) > 0 ? hipErrorOutOfMemory : hipMalloc(&q, need + 256);
      if (e != hipSuccess) return fail(MH_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed: %s", need + 256, hipGetErrorString(e));
      bytes = need + 256;
    } else {
      bytes = cap;
    }
    if (p) {
      if (n_retired == 24) free_retired();
      retired[n_retired++] = p;
    }
    p = q;
    return MH_OK;
  }
  // fault injection (mh_debug_fail_allocations, tests/test_gpu_parity.py): the next N first attempts / retries report
  // out-of-memory without asking the runtime
  static inline std::atomic<int> fault_first_attempts{0}, fault_retries{0};
  void free_retired() {
    for (int i = 0; i < n_retired; i++) (void)hipFree(retired[i]);
    n_retired = 0;
  }
  void release() {
    free_retired();
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  template <class T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

// ---- map layout in HBM -----------------------------------------------------------------------
// One 16-byte slot per hash bucket: a single dwordx4 load answers "is voxel (kx,ky,kz) occupied,
// and where are its points".  key packs 3 x 21-bit biased voxel indices; ~0 = empty.
struct alignas(16) MapSlot {
  unsigned long long key;
  uint32_t first;  // index of the voxel's first point record
  uint32_t count;  // number of point records (<= max_points_per_voxel)
};
static_assert(sizeof(MapSlot) == 16, "slot must be one dwordx4");

constexpr unsigned long long kEmptyKey = ~0ull;
constexpr int kKeyBias = 1 << 20;

struct MapView {
  const MapSlot* slots;
  const float4* pts;  // {x,y,z, bit-cast source index}, voxel-contiguous
  uint32_t mask;      // table_size - 1
  float inv_vs;
  float vs;           // 1.0f / inv_vs, divided once on the host (the kernels only need it for the pruning bounds)
  uint32_t trunc;     // index_mode == MH_INDEX_TRUNC
  uint32_t ndt;       // 1: every voxel's points are preceded by two records {centroid, plane flag} {normal, 0}
  uint32_t no_prev_bound;  // A/B switch (MH_NO_PREV_BOUND=1): the quad matcher ignores the previous iteration's pairing
  // sub-voxel index of the quad matcher (round 4; valid after map_ensure_qidx, null before): pts_q = the records of every voxel
  // re-ordered by (x half, y half) of the voxel, w = the record's index in `pts` (the reference's scan position: the
  // tie-break); the quadrants' boundaries ride in the count word of the voxel's hash slot (slot_count(), mh_nn_device.h)
  const float4* pts_q;
#ifdef MH_DEBUG_WAVETRACE
  uint32_t dbg_stop;  // debug build: leave the quad search after phase N (tools/wavetrace_probe.py)
#endif
};

__host__ __device__ inline unsigned long long pack_key(int kx, int ky, int kz) {
  return ((unsigned long long)(unsigned)(kx + kKeyBias) << 42) | ((unsigned long long)(unsigned)(ky + kKeyBias) << 21) |
         (unsigned long long)(unsigned)(kz + kKeyBias);
}
__host__ __device__ inline void unpack_key(unsigned long long k, int& kx, int& ky, int& kz) {
  kx = (int)((k >> 42) & 0x1FFFFF) - kKeyBias;
  ky = (int)((k >> 21) & 0x1FFFFF) - kKeyBias;
  kz = (int)(k & 0x1FFFFF) - kKeyBias;
}
__host__ __device__ inline uint32_t hash_key(unsigned long long k) {
  return (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> 32);  // Fibonacci hashing of the packed key
}
__host__ __device__ inline bool key_in_range(int k) { return k > -kKeyBias + 1 && k < kKeyBias - 2; }

}  // namespace mh

// ---- opaque handle bodies ---------------------------------------------------------------------
struct IcpDeviceState;   // mh_icp.hip
struct IcpDeviceParams;  // mh_icp.hip

struct mh_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // scratch (grow-only), all on `device`
  mh::DevBuf pair_q;      // float4 per scan point: NN point xyz + d2
  mh::DevBuf pair_gidx;   // uint32 per scan point: source index or 0xFFFFFFFF
  mh::DevBuf pl_c, pl_n;  // float4 per scan point: point-to-plane pairing {centroid, valid flag} {normal, 0}
  mh::DevBuf partials;    // per-block reduction partials (double)
  mh::DevBuf partials_b;  // generic (pt2pl) partials
  mh::DevBuf loop_x;      // k_icp16: the exchange block of the one-launch loop (16-byte entries, two halves)
  uint32_t loop_serial = 0;  // ... and the serial number its next loop's entries start from (never repeats)
  mh::DevBuf sched;       // threshold / kernel-param arrays (double)
  mh::DevBuf batch_desc, batch_states;  // lock-step batches led by this context: job descriptors, gathered states
  void* h_batch = nullptr;              // pinned mirror of both
  size_t h_batch_cap = 0;
  // iterations the last auto-chunked alignment needed (sizes the next first chunk when the caller gives no estimate)
  uint32_t predicted_iterations[2] = {0, 0};
  double* h_sched = nullptr;  // pinned staging for them
  size_t h_sched_cap = 0;
  mh::DevBuf trace;       // mh_icp_iter[max_iterations]
  mh::DevBuf compact;     // compaction scratch (block counts / offsets) and staged outputs
  mh::DevBuf staging;     // generic staging for host<->device array transfers
  mh::DevBuf sort_tmp;    // rocprim temporary storage
  mh::DevBuf build_a, build_b, build_c, build_d, build_e;  // map build scratch
  IcpDeviceState* d_state = nullptr;
  IcpDeviceState* d_state_b = nullptr;  // k_step16: the other half of the state ping-pong (same allocation, head only)
  IcpDeviceState* h_state = nullptr;  // pinned mirror; d_state/h_state own one block [state | params]
  IcpDeviceParams* d_params = nullptr;  // per-alignment parameters (kernels take pointers into this block)
  IcpDeviceParams* h_params = nullptr;  // pinned mirror
  uint32_t* h_progress = nullptr;  // page-locked word behind the state block: (iteration | done << 31), written by the device loop
  uint32_t* d_progress = nullptr;  // ... its device-visible address
  hipGraphExec_t graph_exec = nullptr;  // captured chunk of ICP iterations (replayed while graph_key matches)
  unsigned long long graph_key[32] = {0};
  unsigned long long graph_candidate[32] = {0};  // key of the last direct-launched chunk: captured when a LATER alignment repeats it
  unsigned long long graph_candidate_align = 0, align_serial = 0;
  uint32_t* h_small = nullptr;  // pinned, device-visible [64]: small results a kernel writes straight to the host (mh_scan_bbox)
  char* h_pp = nullptr;      // page-locked staging of the filter chain: job descriptors up, counters down
  size_t h_pp_bytes = 0;
  hipEvent_t ev_poll = nullptr;
  hipEvent_t ev_ready = nullptr;  // "everything queued on this context's stream so far": what a batch leader waits for
  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
  // batches led by this context: second stream for the download of the final pairings (overlaps the next batch)
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_pairs_ready = nullptr, ev_pairs_copied = nullptr;
  bool pairs_copy_pending = false;
  mh::DevBuf pairs_stage;  // compacted pairings of all jobs of a batch
  // profiling events for the match kernel (pairs), created lazily
  hipEvent_t* prof_ev = nullptr;
  uint32_t prof_cap = 0;
};

struct mh_map {
  mh_ctx* ctx = nullptr;
  mh_map_params params{};
  float inv_vs = 1.f;
  mh::DevBuf slots;      // MapSlot[table_size]
  mh::DevBuf pts;        // float4[n_points]
  // sub-voxel index of the quad matcher (MapView::pts_q + the boundaries in the slots' count words), built lazily by map_ensure_qidx after every (re)build
  mh::DevBuf pts_q;
  std::mutex qidx_mtx;
  std::atomic<bool> qidx_valid{false};  // (read by view() without the mutex)
  bool qidx_pending = false;
  hipEvent_t ev_qidx = nullptr;
  hipStream_t qidx_stream = nullptr;
  uint32_t* d_counters = nullptr;  // the device counters of the last (re)build ([9] = voxels)
  mh::DevBuf vox_keys;   // uint64[n_voxels], ascending
  mh::DevBuf vox_first;  // uint32[n_voxels]
  mh::DevBuf vox_count;  // uint32[n_voxels]
  // Counts and bounding box of the stored content.  A (re)build leaves them on the device and copies them into the pinned
  // block `h_counts` asynchronously; they are read back lazily (mh::map_resolve) by whoever needs them on the host -- the
  // next insertion, mh_map_get_info, a download -- so that a key-frame update costs no host synchronisation.
  mutable uint64_t n_points = 0, n_voxels = 0, n_records = 0, n_planes = 0;
  uint64_t n_offered = 0, table_size = 0;
  mh::DevBuf merge;  // mh_map_insert staging: x | y | z | src of (stored + new) points
  mutable float bbox_min[3] = {0, 0, 0}, bbox_max[3] = {0, 0, 0};
  uint32_t* h_counts = nullptr;          // pinned [16]: what k_scatter / k_ndt_stats left in the device counters
  hipEvent_t ev_counts = nullptr;        // recorded behind that copy: "the last (re)build is complete"
  mutable bool counts_pending = false;   // n_points ... bbox above are stale until map_resolve() has seen ev_counts
  mutable bool build_in_flight = false;  // the last (re)build may still run: consumers on OTHER streams order themselves with map_ready_on()
  mutable mh_status deferred_error = MH_OK;  // an insertion's out-of-range verdict, reported by the next call that resolves
  // mh_map_insert's scratch is the map's own
  mh::DevBuf build_a, build_b, build_c, build_d, build_e, sort_tmp;
  mh::MapView view() const {
    mh::MapView v;
    v.slots = slots.as<mh::MapSlot>();
    v.pts = pts.as<float4>();
    v.mask = (uint32_t)(table_size ? table_size - 1 : 0);
    v.inv_vs = inv_vs;
    v.vs = 1.0f / inv_vs;
    v.trunc = params.index_mode == MH_INDEX_TRUNC;
    v.ndt = params.ndt_max_eigen_ratio > 0.f ? 1u : 0u;
    v.no_prev_bound = getenv("MH_NO_PREV_BOUND") != nullptr ? 1u : 0u;
    v.pts_q = qidx_valid ? pts_q.as<float4>() : nullptr;
#ifdef MH_DEBUG_WAVETRACE
    v.dbg_stop = getenv("MH_DBG_STOP") ? (uint32_t)atoi(getenv("MH_DBG_STOP")) : 0u;
#endif
    return v;
  }
};

struct mh_scan {
  mh_ctx* ctx = nullptr;
  mh::DevBuf xyz;  // own storage: x[n] | y[n] | z[n] (SoA, 256-byte aligned sections)
  mh::DevBuf aux;  // optional channels: t[n] | src[n]
  const float *x = nullptr, *y = nullptr, *z = nullptr;  // device pointers actually used
  const float* t = nullptr;       // per-point time stamps [s] or null
  const uint32_t* src = nullptr;  // index of each point in the raw scan it was filtered from, or null
  size_t n = 0;
  // Search order of the tile matcher (mh_tile.hip), a cache that belongs to the point content: the points sorted by
  // (2x2x2-voxel block of the LOCAL frame, quarter-voxel Morton code inside it) and cut into tiles of <= 256 consecutive
  // points that never cross a block.  A rigid transform keeps a tile's points together, so the map records one tile needs
  // fit one workgroup's LDS whatever the pose.  Built lazily (scan_build_tiles), dropped whenever the points change.
  mutable mh::DevBuf tiles;  // sx | sy | sz | perm | tile_start[n + 1]
  mutable const float *sx = nullptr, *sy = nullptr, *sz = nullptr;
  mutable const uint32_t* perm = nullptr;        // sorted position -> index in x/y/z
  mutable const uint32_t* tile_start = nullptr;  // [n_tiles + 1]
  mutable uint32_t n_tiles = 0;
  mutable float tile_inv_vs = 0.f;
  mutable uint32_t tile_points = 0;
  mutable bool tiles_valid = false, tiles_pending = false;  // pending: n_tiles still travelling to h_ntiles
  mutable uint32_t* h_ntiles = nullptr;  // pinned
  mutable hipEvent_t ev_tiles = nullptr;
};

namespace mh {
// Blocking waits of the library (one place).
hipError_t wait_stream(hipStream_t s);
hipError_t wait_event(hipEvent_t e);
mh_status set_device(const mh_ctx* ctx);
// copy `n` elements of a caller array living in `mem` into device scratch (returns device ptr)
mh_status stage_in(mh_ctx* ctx, DevBuf& buf, size_t offset_bytes, const void* src, size_t bytes, int32_t mem);
// (re)size a scan's own storage for n points (+ optional t / src channels) and point x,y,z,t,src at it
mh_status scan_alloc(mh_scan* s, size_t n, bool with_t, bool with_src);
// map (re)build from device arrays; src_ids null = identity.  evict: 0 or {cx,cy,cz,dist_in_grid} voxel test.
// n_stored: the first n_stored inputs are points the map already stores (accepted by the insertion rules before).
// sorted copy + tile table of a scan for voxel size 1/inv_vs (no-op when valid); asynchronous on the scan's stream,
// scan_tiles_ready() waits for the tile count
#ifdef MH_DEV_VARIANTS  // (mh_tile.hip: the search order of the tile / wave / sorted matchers -- development library only)
mh_status scan_build_tiles(const mh_scan* s, float inv_vs, uint32_t tile_points);
uint32_t tile_points_for_env();
mh_status scan_tiles_ready(const mh_scan* s);
void scan_drop_tiles(mh_scan* s);  // host-side bookkeeping only (the points changed)
void scan_free_tiles(mh_scan* s);
#else
inline mh_status scan_build_tiles(const mh_scan*, float, uint32_t) { return MH_OK; }
inline uint32_t tile_points_for_env() { return 0; }
inline mh_status scan_tiles_ready(const mh_scan*) { return MH_OK; }
inline void scan_drop_tiles(mh_scan*) {}
inline void scan_free_tiles(mh_scan*) {}
#endif
// Asynchronous on stream `s` (the context's); scratch from `m`.  Counts / bbox / the
// out-of-range verdict are resolved lazily (map_resolve).
mh_status map_build_device(mh_map* m, hipStream_t s, const float* dx, const float* dy, const float* dz, const uint32_t* dsrc,
                           size_t n, const int* evict, size_t n_stored, bool collected = false);
mh_status map_build_prologue(mh_map* m, hipStream_t s, size_t n, size_t n_stored, uint32_t** counters_out,
                             unsigned long long** keys_out, uint32_t** idx_out);
// wait (host) for the last (re)build's counters and refresh n_points / n_voxels / n_records / n_planes / bbox; returns the
// deferred status of that build (MH_ERR_OUT_OF_RANGE) once
mh_status map_resolve(const mh_map* m);
// the same without handing the verdict out: counts, bounding box; the verdict stays recorded in m->deferred_error
mh_status map_resolve_counts(const mh_map* m);
// make stream `s` wait for a (re)build that may still be running on the map's own context stream (no-op when `s` is that stream)
mh_status map_ready_on(const mh_map* m, hipStream_t s);
mh_status map_ensure_qidx(const mh_map* m, hipStream_t s);  // before any kernel that runs nn_search_quad (after map_ready_on)
}  // namespace mh
