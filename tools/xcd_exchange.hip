// What does one all-to-all hand-over of partial sums cost when the workgroups that exchange sit on ONE XCD (shared L2) instead of
// all over the part?  The small-layer ICP chain (k_step16) pays ~4.5 us of launch boundary + ~2 us of agent-scope exchange per
// Gauss-Newton step; the question is what a persistent loop confined to one XCD would pay instead.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/xcd_exchange tools/xcd_exchange.hip && tools/xcd_exchange
//
// A launch of 256 workgroups x 512 threads (100 KB of LDS each: one per CU).  Every workgroup reads its XCC id; in the
// "one XCD" modes those not on the target XCD leave at once, the others claim a slot.  Once all have reported, the W
// claimants run `steps` rounds of: store a column of 18 doubles + tag, wait for all W tags, load all W columns, sum.
//   mode 0: one XCD, plain stores (write-through to the XCD's L2), sc0 loads (L1 bypass, L2 hit)
//   mode 1: one XCD, agent-scope (sc1) stores and loads -- what crossing XCDs costs, same placement
//   mode 2: first 32 claimants wherever they run, agent scope -- the grid-barrier loop that was measured in round 4
// Also prints the XCC id of the first workgroups (is the placement round-robin?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

constexpr int kRows = 18, kMaxW = 64, kThreads = 512;

struct Ctl {
  unsigned claim, arrived, timeouts, pad;
  unsigned long long t0, t1;
  unsigned xcc[256];
  double check[kMaxW];
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), (short)0, (int)bytes, 0x00020000);
}
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
template <int AUX>
__device__ __forceinline__ double ld(__amdgpu_buffer_rsrc_t r, unsigned idx) {
  const u32x2v w = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(idx * 8u), 0, AUX);
  return __hiloint2double((int)w.y, (int)w.x);
}
template <int AUX>
__device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t r, unsigned idx, double v) {
  const u32x2v w = {(unsigned)__double2loint(v), (unsigned)__double2hiint(v)};
  __builtin_amdgcn_raw_buffer_store_b64(w, r, (int)(idx * 8u), 0, AUX);
}

template <int LD_AUX, int ST_AUX>
__device__ void rounds(Ctl* c, double* cols, unsigned slot, unsigned W, unsigned steps) {
  __shared__ double red[kRows][kMaxW];
  __shared__ double tot[kRows];
  __shared__ unsigned gave_up;
  const unsigned tid = threadIdx.x;
  if (tid == 0) gave_up = 0;
  __syncthreads();
  const unsigned half = (kRows + 1) * kMaxW;  // doubles per ping-pong half: rows x columns + a row of tags
  const __amdgpu_buffer_rsrc_t r = rsrc_of(cols, 2u * half * 8u);
  double carry = 1.0;
  if (slot == 0 && tid == 0) c->t0 = wall_clock64();
  for (unsigned s = 0; s < steps; s++) {
    const unsigned out = (s & 1u) * half;
    if (tid < kRows) st<ST_AUX>(r, out + tid * kMaxW + slot, carry + (double)(tid + slot));
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the column is acknowledged
    __syncthreads();
    if (tid == 0) st<ST_AUX>(r, out + kRows * kMaxW + slot, (double)(s + 1u));
    // wait for every column's tag
    if (tid < W) {
      unsigned spins = 0;
      while (ld<LD_AUX>(r, out + kRows * kMaxW + tid) != (double)(s + 1u)) {
        if (++spins > (1u << 14) || __hip_atomic_load(&c->timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {  // (somebody gave up: everybody does)
          atomicAdd(&c->timeouts, 1u);
          gave_up = 1u;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (gave_up) break;
    // all rows x columns: one load per lane (18 x W <= 512 for W <= 28; two rounds beyond)
    for (unsigned e = tid; e < kRows * W; e += kThreads) {
      const unsigned row = e / W, col = e % W;
      red[row][col] = ld<LD_AUX>(r, out + row * kMaxW + col);
    }
    __syncthreads();
    if (tid < kRows) {
      double acc = 0.0;
      for (unsigned q = 0; q < W; q++) acc += red[tid][q];
      tot[tid] = acc;
    }
    __syncthreads();
    carry = tot[0] * 1e-3 + 1.0;  // (the next round depends on this one)
  }
  if (slot == 0 && tid == 0) c->t1 = wall_clock64();
  if (tid == 0) c->check[slot] = carry;
}

// mode 3: every entry is 16 bytes {value, serial | check word}, one sc1 store / load each: no separate tag, ONE round trip; a
// reader retries the entries that do not carry the round's serial number yet.  `rows` doubles per column.
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
__device__ void rounds_fused(Ctl* c, double* cols, unsigned slot, unsigned W, unsigned steps, unsigned rows, unsigned rs) {
  __shared__ double red[48][kMaxW];
  __shared__ double tot[48];
  __shared__ unsigned gave_up;
  const unsigned tid = threadIdx.x;
  if (tid == 0) gave_up = 0;
  __syncthreads();
  const unsigned half = 48u * rs;  // entries per ping-pong half (rs: entries from one row to the next)
  const __amdgpu_buffer_rsrc_t r = rsrc_of(cols, 2u * half * 16u);
  double carry = 1.0;
  if (slot == 0 && tid == 0) c->t0 = wall_clock64();
  for (unsigned s = 0; s < steps; s++) {
    const unsigned out = (s & 1u) * half;
    if (tid < rows) {
      const double v = carry + (double)(tid + slot);
      const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
      const u32x4v w = {lo, hi, s + 1u, lo ^ hi ^ (s + 1u)};
      __builtin_amdgcn_raw_buffer_store_b128(w, r, (int)((out + tid * rs + slot) * 16u), 0, 16);
    }
    for (unsigned e = tid; e < rows * W; e += kThreads) {
      const unsigned row = e / W, col = e % W;
      unsigned spins = 0;
      u32x4v w;
      for (;;) {
        w = __builtin_amdgcn_raw_buffer_load_b128(r, (int)((out + row * rs + col) * 16u), 0, 16);
        if (w.z == s + 1u && w.w == (w.x ^ w.y ^ w.z)) break;
        if (++spins > (1u << 14) || __hip_atomic_load(&c->timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          atomicAdd(&c->timeouts, 1u);
          gave_up = 1u;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      red[row][col] = __hiloint2double((int)w.y, (int)w.x);
    }
    __syncthreads();
    if (gave_up) break;
    if (tid < rows) {
      double acc = 0.0;
      for (unsigned q = 0; q < W; q++) acc += red[tid][q];
      tot[tid] = acc;
    }
    __syncthreads();
    carry = tot[0] * 1e-3 + 1.0;
  }
  if (slot == 0 && tid == 0) c->t1 = wall_clock64();
  if (tid == 0) c->check[slot] = carry;
}

__global__ __launch_bounds__(kThreads) void k_xchg(Ctl* c, double* cols, unsigned mode, unsigned target, unsigned steps, unsigned wmax, unsigned rows, unsigned rs) {
  extern __shared__ char pad_[];  // (one workgroup per CU)
  (void)pad_;
  __shared__ unsigned s_slot, s_w;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 15u;
  const unsigned tid = threadIdx.x;
  if (tid == 0) {
    if (blockIdx.x < 256) c->xcc[blockIdx.x] = xcc;
    unsigned slot = ~0u;
    if (mode >= 2 || xcc == target) {
      slot = __hip_atomic_fetch_add(&c->claim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (slot >= wmax) slot = ~0u;
    }
    __hip_atomic_fetch_add(&c->arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    s_slot = slot;
    if (slot != ~0u) {
      unsigned spins = 0;
      while (__hip_atomic_load(&c->arrived, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
        if (++spins > (1u << 22)) {
          atomicAdd(&c->timeouts, 1u << 16);
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
      const unsigned cl = __hip_atomic_load(&c->claim, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_w = cl < wmax ? cl : wmax;
    }
  }
  __syncthreads();
  const unsigned slot = s_slot;
  if (slot == ~0u) return;
  const unsigned W = s_w;
  if (mode == 3) rounds_fused(c, cols, slot, W, steps, rows, rs);
  else if (mode == 0) rounds<1, 0>(c, cols, slot, W, steps);
  else rounds<16, 16>(c, cols, slot, W, steps);
}

int main(int argc, char** argv) {
  const unsigned steps = argc > 1 ? (unsigned)atoi(argv[1]) : 2000u;
  Ctl* c;
  double* cols;
  CK(hipMalloc(&c, sizeof(Ctl)));
  const unsigned rs = getenv("ROW_STRIDE") ? (unsigned)atoi(getenv("ROW_STRIDE")) : (unsigned)kMaxW;
  const size_t cols_bytes = (size_t)2 * 48 * (rs > (unsigned)kMaxW ? rs : (unsigned)kMaxW) * 16;
  CK(hipMalloc(&cols, cols_bytes));
  int clk_khz = 0;
  CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0));
  const char* names[5] = {"one XCD, stores to L2 + sc0 loads", "one XCD, agent scope (sc1)", "anywhere, agent scope (sc1), tags", "anywhere, 16-byte entries, 18 rows", "anywhere, 16-byte entries, 47 rows"};
  for (unsigned wmax : {64u, 44u, 32u, 16u}) {
    for (unsigned mode = 0; mode < 5; mode++) {
      for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(c, 0, sizeof(Ctl)));
        CK(hipMemset(cols, 0, cols_bytes));
        hipLaunchKernelGGL(k_xchg, dim3(256), dim3(kThreads), 100 * 1024, 0, c, cols, mode < 4 ? mode : 3u, 3u, steps, wmax, mode == 4 ? 47u : 18u, rs);
        CK(hipDeviceSynchronize());
        Ctl h;
        CK(hipMemcpy(&h, c, sizeof(h), hipMemcpyDeviceToHost));
        if (rep == 0 && mode == 0 && wmax == 64u) {
          printf("xcc of workgroups 0..31:");
          for (int i = 0; i < 32; i++) printf(" %u", h.xcc[i]);
          int mism = 0, hist[16] = {0};
          for (int i = 0; i < 256; i++) {
            hist[h.xcc[i] & 15]++;
            if (h.xcc[i] != (unsigned)(i % 8)) mism++;
          }
          fflush(stdout);
          printf("\nworkgroups whose XCC id is not blockIdx %% 8: %d of 256; per XCC:", mism);
          for (int i = 0; i < 8; i++) printf(" %d", hist[i]);
          printf("\n");
        }
        if (rep == 1) {
          const unsigned w = h.claim < wmax ? h.claim : wmax;
          const double us = (double)(h.t1 - h.t0) / (double)clk_khz * 1e3 / steps;
          bool same = true;
          for (unsigned i = 1; i < w; i++) same = same && h.check[i] == h.check[0];
          printf("W<=%2u  %-44s claimants %3u  workers %2u  %.3f us per round  timeouts %u  all agree %d  (carry %.6f)\n", wmax, names[mode],
                 h.claim, w, us, h.timeouts, (int)same, h.check[0]);
          fflush(stdout);
        }
      }
    }
  }
  return 0;
}
