// Calibration of rocprofv3's FETCH_SIZE on MI355X for the access patterns of the match kernel:
// kernels that move a KNOWN number of bytes from a table far larger than L2 + Infinity Cache, so every request reaches
// the memory side.  Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (tools/fetch_calib.sh) and compare the counter
// with the byte counts printed here.  Build: hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__host__ __device__ inline unsigned hash32(unsigned x){ x^=x>>16; x*=0x7feb352dU; x^=x>>15; x*=0x846ca68bU; x^=x>>16; return x; }

// every lane one 16-byte load; GROUP consecutive lanes read GROUP consecutive records (GROUP*16 contiguous bytes) at a
// random, GROUP-aligned place of the table: GROUP = 64 -> a wave streams 1 KiB, 4 -> the quad pattern of k_match4
// (64 B), 1 -> lone 16-byte gathers (the hash-slot probes)
template <int GROUP>
__global__ void k_calib(const u32x4* __restrict__ tab, unsigned long long mask, unsigned n, unsigned salt, unsigned* __restrict__ out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned g = i / GROUP;
  const unsigned long long base = ((((unsigned long long)hash32(g * 2u + salt) << 32) | hash32(g * 2u + 1u + salt)) & mask) & ~(unsigned long long)(GROUP - 1);
  const u32x4 v = tab[base + (i % GROUP)];
  if (v.x == 0xdeadbeefu) out[0] = v.y;  // (never true: keeps the load)
}
// plain streaming read of n*16 bytes
__global__ void k_stream(const u32x4* __restrict__ tab, unsigned n, unsigned* __restrict__ out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32x4 v = tab[i];
  if (v.x == 0xdeadbeefu) out[0] = v.y;
}

int main() {
  const unsigned long long REC = 1ull << 27;  // 2^27 records x 16 B = 2 GiB  (L2 4 MiB/XCD, Infinity Cache 256 MiB)
  u32x4* tab; unsigned* out;
  CK(hipMalloc(&tab, REC * 16)); CK(hipMalloc(&out, 64));
  CK(hipMemset(tab, 1, REC * 16)); CK(hipMemset(out, 0, 64));
  const unsigned n = 1u << 24;  // 16 M loads per launch = 256 MiB requested
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 3; rep++) {  // three launches each; different places every time (no reuse across launches)
    hipLaunchKernelGGL(k_stream, dim3(n / 256), dim3(256), 0, 0, tab + (size_t)rep * n, n, out);
    hipLaunchKernelGGL(k_calib<64>, dim3(n / 256), dim3(256), 0, 0, tab, REC - 1, n, 1000u * rep + 1, out);
    hipLaunchKernelGGL(k_calib<4>, dim3(n / 256), dim3(256), 0, 0, tab, REC - 1, n, 1000u * rep + 2, out);
    hipLaunchKernelGGL(k_calib<1>, dim3(n / 256), dim3(256), 0, 0, tab, REC - 1, n, 1000u * rep + 3, out);
  }
  CK(hipDeviceSynchronize());
  printf("{\"loads_per_launch\": %u, \"requested_bytes_per_launch\": %llu, \"table_bytes\": %llu, "
         "\"distinct_64B_blocks\": {\"k_stream\": %u, \"k_calib<64>\": %u, \"k_calib<4>\": %u, \"k_calib<1>\": %u}}\n",
         n, (unsigned long long)n * 16, REC * 16, n / 4, n / 4, n / 4, n);
  return 0;
}
