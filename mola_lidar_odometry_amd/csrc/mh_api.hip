// mh_api.hip -- library, context and scan entry points of the C ABI (include/molahip.h).
#include <stdarg.h>
#include <string.h>

#include <new>

#include "mh_internal.h"

namespace mh {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

mh_status fail(mh_status s, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return s;
}

// Blocking waits of the library, in one place.  (Round 3 could turn them into "mark the stream, call a hook until done" for
// callers that ran several sequences as fibers of one host thread -- mh_set_wait_hook, molahip-lo-cli --fibers: slower than a
// thread per sequence at every count, 2410-2650 against 4580-4920 scans/s for eight; removed in round 4.)
hipError_t wait_event(hipEvent_t e) { return hipEventSynchronize(e); }
hipError_t wait_stream(hipStream_t s) { return hipStreamSynchronize(s); }

mh_status set_device(const mh_ctx* ctx) {
  MH_HIP(hipSetDevice(ctx->device));
  return MH_OK;
}

mh_status stage_in(mh_ctx* ctx, DevBuf& buf, size_t offset_bytes, const void* src, size_t bytes, int32_t mem) {
  if (bytes == 0) return MH_OK;
  char* dst = buf.as<char>() + offset_bytes;
  MH_HIP(hipMemcpyAsync(dst, src, bytes, mem == MH_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                        ctx->stream));
  return MH_OK;
}

mh_status scan_alloc(mh_scan* s, size_t n, bool with_t, bool with_src) {
  mh_ctx* ctx = s->ctx;
  const size_t stride = ((n * sizeof(float) + 255) / 256) * 256;
  if (s->xyz.bytes < 3 * stride || (s->aux.bytes < 2 * stride && (with_t || with_src))) {
    MH_HIP(mh::wait_stream(ctx->stream));  // nobody may still read the old buffers
    MH_TRY(s->xyz.reserve(3 * stride ? 3 * stride : 256));
    if (with_t || with_src) MH_TRY(s->aux.reserve(2 * stride ? 2 * stride : 256));
  }
  char* base = s->xyz.as<char>();
  s->x = (const float*)base;
  s->y = (const float*)(base + stride);
  s->z = (const float*)(base + 2 * stride);
  s->t = with_t ? (const float*)s->aux.as<char>() : nullptr;
  s->src = with_src ? (const uint32_t*)(s->aux.as<char>() + stride) : nullptr;
  s->n = n;
  scan_drop_tiles(s);
  return MH_OK;
}

}  // namespace mh

using namespace mh;

extern "C" {
uint32_t mh_abi_version(void) { return MH_ABI_VERSION; }


mh_status mh_host_alloc_pinned(size_t bytes, void** out) {
  MH_REQUIRE(out, "null argument");
  *out = nullptr;
  MH_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
  return MH_OK;
}

mh_status mh_host_free_pinned(void* p) {
  if (p) MH_HIP(hipHostFree(p));
  return MH_OK;
}


mh_status mh_version(uint32_t* major, uint32_t* minor, uint32_t* patch) {
  if (major) *major = MH_VERSION_MAJOR;
  if (minor) *minor = MH_VERSION_MINOR;
  if (patch) *patch = MH_VERSION_PATCH;
  return MH_OK;
}

const char* mh_last_error_string(void) { return g_err; }

const char* mh_status_string(mh_status s) {
  switch (s) {
    case MH_OK: return "MH_OK";
    case MH_ERR_INVALID_ARGUMENT: return "MH_ERR_INVALID_ARGUMENT";
    case MH_ERR_HIP: return "MH_ERR_HIP";
    case MH_ERR_OUT_OF_MEMORY: return "MH_ERR_OUT_OF_MEMORY";
    case MH_ERR_OUT_OF_RANGE: return "MH_ERR_OUT_OF_RANGE";
    case MH_ERR_NO_DEVICE: return "MH_ERR_NO_DEVICE";
    case MH_ERR_UNSUPPORTED: return "MH_ERR_UNSUPPORTED";
    case MH_ERR_INTERNAL: return "MH_ERR_INTERNAL";
    default: return "MH_ERR_?";
  }
}

mh_status mh_debug_fail_allocations(int32_t first_attempts, int32_t retries) {
  MH_REQUIRE(first_attempts >= 0 && retries >= 0, "negative count");
  mh::DevBuf::fault_first_attempts.store(first_attempts);
  mh::DevBuf::fault_retries.store(retries);
  return MH_OK;
}

mh_status mh_device_count(int32_t* n) {
  MH_REQUIRE(n, "null output");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *n = 0;
    return fail(MH_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *n = c;
  return MH_OK;
}



// (Round 3 also offered contexts whose stream had a priority class or a hardware CU mask, for the prefetch context beside a
// latency-bound alignment: stream priorities 2180 against 2870 scans/s, a CU mask 4530-4650 against 4780 -- removed in round 4.)
mh_status mh_ctx_create(int32_t device, void* hip_stream, mh_ctx** out) {
  MH_REQUIRE(out, "null output");
  *out = nullptr;
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess || c <= 0)
    return fail(MH_ERR_NO_DEVICE, "no HIP device available (%s); libmolahip has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  if (device < 0 || device >= c) return fail(MH_ERR_INVALID_ARGUMENT, "device %d out of range [0,%d)", device, c);
  MH_HIP(hipSetDevice(device));
  mh_ctx* ctx = new (std::nothrow) mh_ctx();
  if (!ctx) return fail(MH_ERR_OUT_OF_MEMORY, "host allocation failed");
  ctx->device = device;
  if (hip_stream) {
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
  } else {
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete ctx;
      return fail(MH_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    ctx->own_stream = true;
  }
  hipError_t e1 = hipEventCreateWithFlags(&ctx->ev_poll, hipEventDisableTiming);
  hipError_t e2 = hipEventCreate(&ctx->ev_t0);
  hipError_t e3 = hipEventCreate(&ctx->ev_t1);
  if (e1 == hipSuccess) e1 = hipEventCreateWithFlags(&ctx->ev_ready, hipEventDisableTiming);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
    mh_ctx_destroy(ctx);
    return fail(MH_ERR_HIP, "hipEventCreate failed");
  }
  *out = ctx;
  return MH_OK;
}

mh_status mh_ctx_destroy(mh_ctx* ctx) {
  if (!ctx) return MH_OK;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)mh::wait_stream(ctx->stream);
  if (ctx->copy_stream) {
    (void)mh::wait_stream(ctx->copy_stream);
    (void)hipStreamDestroy(ctx->copy_stream);
  }
  if (ctx->ev_ready) (void)hipEventDestroy(ctx->ev_ready);
  if (ctx->ev_pairs_ready) (void)hipEventDestroy(ctx->ev_pairs_ready);
  if (ctx->ev_pairs_copied) (void)hipEventDestroy(ctx->ev_pairs_copied);
  ctx->pairs_stage.release();
  ctx->pair_q.release();
  ctx->pair_gidx.release();
  ctx->pl_c.release();
  ctx->pl_n.release();
  ctx->partials.release();
  ctx->partials_b.release();
  ctx->sched.release();
  ctx->trace.release();
  ctx->compact.release();
  ctx->staging.release();
  ctx->sort_tmp.release();
  ctx->build_a.release();
  ctx->build_b.release();
  ctx->build_c.release();
  ctx->build_d.release();
  ctx->build_e.release();
  if (ctx->graph_exec) (void)hipGraphExecDestroy(ctx->graph_exec);
  if (ctx->d_state) (void)hipFree(ctx->d_state);
  if (ctx->h_state) (void)hipHostFree(ctx->h_state);
  if (ctx->h_small) (void)hipHostFree(ctx->h_small);
  if (ctx->h_pp) (void)hipHostFree(ctx->h_pp);
  if (ctx->h_sched) (void)hipHostFree(ctx->h_sched);
  if (ctx->h_batch) (void)hipHostFree(ctx->h_batch);
  ctx->batch_desc.release();
  ctx->batch_states.release();  // d_params / h_params point into the state blocks
  if (ctx->ev_poll) (void)hipEventDestroy(ctx->ev_poll);
  if (ctx->ev_t0) (void)hipEventDestroy(ctx->ev_t0);
  if (ctx->ev_t1) (void)hipEventDestroy(ctx->ev_t1);
  for (uint32_t i = 0; i < ctx->prof_cap; i++) (void)hipEventDestroy(ctx->prof_ev[i]);
  delete[] ctx->prof_ev;
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return MH_OK;
}

mh_status mh_ctx_synchronize(mh_ctx* ctx) {
  MH_REQUIRE(ctx, "null context");
  MH_TRY(set_device(ctx));
  MH_HIP(mh::wait_stream(ctx->stream));
  if (ctx->copy_stream) {  // a batch led by this context may still be downloading its final pairings
    MH_HIP(mh::wait_stream(ctx->copy_stream));
    ctx->pairs_copy_pending = false;
  }
  return MH_OK;
}

mh_status mh_ctx_memory_info(mh_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes) {
  MH_REQUIRE(ctx && free_bytes && total_bytes, "null argument");
  MH_TRY(set_device(ctx));
  MH_HIP(mh::wait_stream(ctx->stream));
  size_t f = 0, t = 0;
  MH_HIP(hipMemGetInfo(&f, &t));
  *free_bytes = f;
  *total_bytes = t;
  return MH_OK;
}

mh_status mh_ctx_stream(mh_ctx* ctx, void** hip_stream_out) {
  MH_REQUIRE(ctx && hip_stream_out, "null argument");
  *hip_stream_out = (void*)ctx->stream;
  return MH_OK;
}

// ---- scan -----------------------------------------------------------------------------------
static mh_status scan_set(mh_scan* s, const float* x, const float* y, const float* z, size_t n, int32_t mem) {
  mh_ctx* ctx = s->ctx;
  MH_REQUIRE(mem == MH_MEM_HOST || mem == MH_MEM_DEVICE || mem == MH_MEM_HOST_PINNED, "bad mem space");
  MH_REQUIRE(n == 0 || (x && y && z), "null point arrays");
  MH_REQUIRE(n < 0x7FFFFFFFull, "scan too large");
  MH_TRY(set_device(ctx));
  // own SoA copy so that the caller's arrays are only borrowed for the call (SURVEY 8b ownership)
  const size_t stride = ((n * sizeof(float) + 255) / 256) * 256;
  if (s->xyz.bytes < 3 * stride) {
    MH_HIP(mh::wait_stream(ctx->stream));  // nobody may still read the old buffer
    MH_TRY(s->xyz.reserve(3 * stride ? 3 * stride : 256));
  }
  char* base = s->xyz.as<char>();
  const hipMemcpyKind kind = mem == MH_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  if (n) {
    MH_HIP(hipMemcpyAsync(base, x, n * sizeof(float), kind, ctx->stream));
    MH_HIP(hipMemcpyAsync(base + stride, y, n * sizeof(float), kind, ctx->stream));
    MH_HIP(hipMemcpyAsync(base + 2 * stride, z, n * sizeof(float), kind, ctx->stream));
    if (mem == MH_MEM_HOST) MH_HIP(mh::wait_stream(ctx->stream));  // host arrays are borrowed only for the call
    // (MH_MEM_HOST_PINNED: the caller keeps them valid until the stream has passed the copies; no wait here)
  }
  s->x = (const float*)base;
  s->y = (const float*)(base + stride);
  s->z = (const float*)(base + 2 * stride);
  s->t = nullptr;  // new points: whatever channels the old ones carried are gone
  s->src = nullptr;
  s->n = n;
  scan_drop_tiles(s);
  return MH_OK;
}

mh_status mh_scan_create(mh_ctx* ctx, const float* x, const float* y, const float* z, size_t n, int32_t mem,
                         mh_scan** out) {
  MH_REQUIRE(ctx && out, "null argument");
  *out = nullptr;
  mh_scan* s = new (std::nothrow) mh_scan();
  if (!s) return fail(MH_ERR_OUT_OF_MEMORY, "host allocation failed");
  s->ctx = ctx;
  mh_status st = scan_set(s, x, y, z, n, mem);
  if (st != MH_OK) {
    s->xyz.release();
    s->aux.release();
    scan_free_tiles(s);
    delete s;
    return st;
  }
  *out = s;
  return MH_OK;
}

mh_status mh_scan_update(mh_scan* scan, const float* x, const float* y, const float* z, size_t n, int32_t mem) {
  MH_REQUIRE(scan, "null scan");
  return scan_set(scan, x, y, z, n, mem);
}

mh_status mh_scan_destroy(mh_scan* scan) {
  if (!scan) return MH_OK;
  (void)hipSetDevice(scan->ctx->device);
  (void)mh::wait_stream(scan->ctx->stream);
  scan->xyz.release();
  scan->aux.release();
  scan_free_tiles(scan);
  delete scan;
  return MH_OK;
}

mh_status mh_scan_size(const mh_scan* scan, uint64_t* n) {
  MH_REQUIRE(scan && n, "null argument");
  *n = scan->n;
  return MH_OK;
}

#ifndef MH_DEV_VARIANTS
// The search order this call queues is what the tile / wave / sorted matchers read (mh_tile.hip: development library only); the
// shipped matchers search the scan as it is, so there is nothing to prepare.
mh_status mh_scan_prepare(const mh_scan* scan, float voxel_size) {
  MH_REQUIRE(scan, "null scan");
  MH_REQUIRE(voxel_size > 0.f, "voxel_size must be > 0");
  return MH_OK;
}
#endif

}  // extern "C"
