// Development microbenchmark (not part of the product): what do dependent / independent 16-byte
// gathers cost on MI355X at the occupancy one 120k-point scan gives?  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

__host__ __device__ inline unsigned hash32(unsigned x){ x^=x>>16; x*=0x7feb352dU; x^=x>>15; x*=0x846ca68bU; x^=x>>16; return x; }

// MODE 0: K independent random loads per lane; 1: K dependent (pointer-chase) loads; 2: K independent loads where
// the 64 lanes of a wave hit only 8 distinct 128-B lines (coherent); 3: K independent, lanes in groups of 8 read 8
// consecutive 16-B records (2 lines per 16 lanes)
template<int MODE, int K>
__global__ void gather(const u32x4* __restrict__ tab, unsigned mask, unsigned n, unsigned* __restrict__ out){
  unsigned i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i>=n) return;
  unsigned acc=0;
  if (MODE==1){
    unsigned idx = hash32(i)&mask;
    for (int k=0;k<K;k++){ u32x4 v = tab[idx]; acc+=v.y; idx = (v.x + k) & mask; }
  } else {
    u32x4 v[K];
    #pragma unroll
    for (int k=0;k<K;k++){
      unsigned idx;
      if (MODE==0) idx = hash32(i*K+k)&mask;
      else if (MODE==2) idx = (hash32((i>>6)*K+k)&mask&~63u) + (i&7)*8 + ((i>>3)&7);
      else if (MODE==3) idx = ((hash32((i>>3)*K+k)&mask)&~7u) + (i&7);
      else if (MODE==4) idx = ((hash32((i>>2)*K+k)&mask)&~3u) + (i&3);      // quads read 4 consecutive records (64 B)
      else if (MODE==5) idx = ((hash32((i>>4)*K+k)&mask)&~15u) + (i&15);    // 16 lanes read 16 consecutive records (256 B)
      else if (MODE==6) idx = ((hash32((i>>6)*K+k)&mask)&~63u) + (i&63);    // the wave reads 64 consecutive records (1 KiB)
      else idx = hash32((i>>6)*K+k)&mask;                                   // the whole wave reads ONE record (broadcast)
      v[k]=tab[idx];
    }
    #pragma unroll
    for (int k=0;k<K;k++) acc+=v[k].x+v[k].w;
  }
  out[i]=acc;
}

template<int MODE,int K> float run(const u32x4* tab, unsigned mask, unsigned n, unsigned* out, const char* name){
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for(int w=0;w<3;w++) hipLaunchKernelGGL((gather<MODE,K>), dim3((n+255)/256), dim3(256),0,0,tab,mask,n,out);
  CK(hipEventRecord(a));
  const int R=20;
  for(int r=0;r<R;r++) hipLaunchKernelGGL((gather<MODE,K>), dim3((n+255)/256), dim3(256),0,0,tab,mask,n,out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms,a,b));
  double us = ms*1e3/R;
  printf("%-34s n=%7u K=%2d : %8.2f us/launch  %7.1f Mloads/ms  %6.2f TB/s(16B)\n", name, n, K, us, (double)n*K/us/1e3, (double)n*K*16/us/1e6);
  return us;
}

int main(){
  const unsigned TAB = 1u<<18; // 262144 slots x 16 B = 4 MiB (the C2 hash table)
  std::vector<unsigned> h(TAB*4);
  for (unsigned i=0;i<TAB;i++){ h[4*i]=hash32(i*7+1); h[4*i+1]=i; h[4*i+2]=i*3; h[4*i+3]=1; }
  u32x4* tab; unsigned* out; CK(hipMalloc(&tab, TAB*16)); CK(hipMalloc(&out, 4u<<20));
  CK(hipMemcpy(tab,h.data(),TAB*16,hipMemcpyHostToDevice));
  // also a 16 MiB table (the C2 point records)
  const unsigned TAB2 = 1u<<20; u32x4* tab2; CK(hipMalloc(&tab2, (size_t)TAB2*16)); CK(hipMemset(tab2, 1, (size_t)TAB2*16));
  unsigned n=120000;
  run<0,1>(tab,TAB-1,n,out,"empty-ish (1 random load)");
  run<0,27>(tab,TAB-1,n,out,"27 independent random, 4MiB");
  run<0,9>(tab,TAB-1,n,out,"9 independent random, 4MiB");
  run<1,27>(tab,TAB-1,n,out,"27 dependent random, 4MiB");
  run<1,9>(tab,TAB-1,n,out,"9 dependent random, 4MiB");
  run<2,27>(tab,TAB-1,n,out,"27 indep, wave-coherent lines");
  run<3,27>(tab,TAB-1,n,out,"27 indep, 8-lane contiguous");
  run<0,27>(tab2,TAB2-1,n,out,"27 independent random, 16MiB");
  run<1,27>(tab2,TAB2-1,n,out,"27 dependent random, 16MiB");
  run<0,27>(tab,TAB-1,n*8,out,"27 indep random, 4MiB, 8x lanes");
  run<1,27>(tab,TAB-1,n*8,out,"27 dependent, 4MiB, 8x lanes");
  run<3,27>(tab,TAB-1,n*8,out,"27 indep 8-lane contig, 8x lanes");
  run<4,27>(tab,TAB-1,n*8,out,"27 indep quad contig, 8x lanes");
  run<5,27>(tab,TAB-1,n*8,out,"27 indep 16-lane contig, 8x lanes");
  run<6,27>(tab,TAB-1,n*8,out,"27 indep wave contig, 8x lanes");
  run<7,27>(tab,TAB-1,n*8,out,"27 indep wave broadcast, 8x lanes");
  run<2,27>(tab,TAB-1,n*8,out,"27 indep wave-coherent, 8x lanes");
  run<0,27>(tab2,TAB2-1,n*8,out,"27 indep random 16MiB, 8x lanes");
  run<4,27>(tab2,TAB2-1,n*8,out,"27 quad contig 16MiB, 8x lanes");
  run<5,27>(tab2,TAB2-1,n*8,out,"27 16-lane contig 16MiB, 8x");
  return 0;
}
