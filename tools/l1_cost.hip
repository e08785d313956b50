// Development microbenchmark (not product): what one vector-memory instruction costs the TA / L1 / TD path of a CU on MI355X
// as a function of HOW the 64 lanes' addresses are arranged and how wide the load is -- the cost model behind the record
// layout of the plan / scan matcher (round 5).  Tables small enough to hit in the 32 KiB L1 (8 KiB) or only in L2 (2 MiB).
// Reports cycles per wave-instruction and CU at full occupancy.  Build: hipcc --offload-arch=gfx950 -O3 -o l1_cost l1_cost.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__host__ __device__ inline unsigned hash32(unsigned x){ x^=x>>16; x*=0x7feb352dU; x^=x>>15; x*=0x846ca68bU; x^=x>>16; return x; }

// G = lanes per contiguous group (1, 4, 8, 16, 64; 0 = broadcast), OFF = misalignment of a group's start in records
template<int G, int OFF, int WIDTH /*dwords per lane: 1, 2, 4*/, int K>
__global__ void gather(const unsigned* __restrict__ tab, unsigned mask /*records - 1*/, unsigned iters, unsigned* __restrict__ out){
  const unsigned i = blockIdx.x*blockDim.x + threadIdx.x;
  unsigned acc = 0;
  for (unsigned it = 0; it < iters; it++) {
    unsigned idx[K];
#pragma unroll
    for (int k=0;k<K;k++){
      const unsigned seed = it*K+k;
      if (G==0) idx[k] = hash32((i>>6)*977u+seed)&mask;
      else if (G==1) idx[k] = hash32(i*977u+seed)&mask;
      else idx[k] = (((hash32((i/G)*977u+seed)&mask)&~(unsigned)(G-1)) + OFF + (i%G)) & mask;
    }
    if (WIDTH==4){ u32x4 v[K];
#pragma unroll
      for (int k=0;k<K;k++) v[k] = reinterpret_cast<const u32x4*>(tab)[idx[k]];
#pragma unroll
      for (int k=0;k<K;k++) acc += v[k].x + v[k].w;
    } else if (WIDTH==2){ u32x2 v[K];
#pragma unroll
      for (int k=0;k<K;k++) v[k] = reinterpret_cast<const u32x2*>(tab)[2*idx[k]];
#pragma unroll
      for (int k=0;k<K;k++) acc += v[k].x + v[k].y;
    } else { unsigned v[K];
#pragma unroll
      for (int k=0;k<K;k++) v[k] = tab[4*idx[k]];
#pragma unroll
      for (int k=0;k<K;k++) acc += v[k];
    }
  }
  if (acc == 0x12345678u) out[i] = acc;
}

template<int G,int OFF,int WIDTH> void run(const unsigned* tab, unsigned records, unsigned* out, const char* name, double ghz){
  constexpr int K = 8;
  const unsigned iters = 64, nblk = 256*8*4, nthr = 256;  // 8 waves per SIMD on every CU, 4 rounds
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((gather<G,OFF,WIDTH,K>), dim3(nblk), dim3(nthr),0,0,tab,records-1,iters,out);
  CK(hipEventRecord(a));
  const int R=5;
  for(int r=0;r<R;r++) hipLaunchKernelGGL((gather<G,OFF,WIDTH,K>), dim3(nblk), dim3(nthr),0,0,tab,records-1,iters,out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms,a,b));
  const double us = ms*1e3/R;
  const double winstr = (double)nblk*nthr/64*iters*K;
  printf("%-44s table %7u KiB : %8.1f us  %6.1f cycles per wave-instruction and CU (%.2f GHz assumed)\n", name, records*16/1024, us, us*1e-6*ghz*1e9*256/winstr, ghz);
}

int main(){
  const double ghz = 2.4;
  unsigned* out; CK(hipMalloc(&out, 64u<<20));
  for (unsigned records : {512u, 131072u}) {   // 8 KiB (L1), 2 MiB (L2)
    unsigned* tab; CK(hipMalloc(&tab, (size_t)records*16)); CK(hipMemset(tab, 1, (size_t)records*16));
    run<0,0,4>(tab,records,out,"dwordx4 broadcast (1 record per wave)",ghz);
    run<64,0,4>(tab,records,out,"dwordx4 wave-contiguous (1 KiB)",ghz);
    run<16,0,4>(tab,records,out,"dwordx4 16 lanes contiguous (256 B)",ghz);
    run<8,0,4>(tab,records,out,"dwordx4 8 lanes contiguous (128 B aligned)",ghz);
    run<8,3,4>(tab,records,out,"dwordx4 8 lanes contiguous (+48 B)",ghz);
    run<4,0,4>(tab,records,out,"dwordx4 quads contiguous (64 B aligned)",ghz);
    run<4,2,4>(tab,records,out,"dwordx4 quads contiguous (+32 B)",ghz);
    run<2,0,4>(tab,records,out,"dwordx4 pairs contiguous (32 B)",ghz);
    run<1,0,4>(tab,records,out,"dwordx4 every lane its own record",ghz);
    run<4,0,2>(tab,records,out,"dwordx2 quads (first 8 B of 4 records)",ghz);
    run<1,0,2>(tab,records,out,"dwordx2 every lane its own record",ghz);
    run<4,0,1>(tab,records,out,"dword quads (first 4 B of 4 records)",ghz);
    run<1,0,1>(tab,records,out,"dword every lane its own record",ghz);
    run<64,0,1>(tab,records,out,"dword wave, stride 16 B",ghz);
    CK(hipFree(tab));
  }
  return 0;
}
