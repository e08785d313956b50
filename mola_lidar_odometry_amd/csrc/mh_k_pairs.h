// mh_k_pairs.h -- Results::finalPairings [U]: ballot / prefix-sum compaction of the pairing buffers in ascending local index.
#pragma once

// ================================================================================================
// Pairing compaction: per-block count -> scan of block counts -> ballot/prefix scatter.
// Output order = ascending local index (what a serial matcher emits).
// ================================================================================================
__device__ __forceinline__ void k_count_valid_body(const uint32_t* __restrict__ gidx, uint32_t n,
                                                   uint32_t* __restrict__ block_counts) {
  __shared__ uint32_t wc[kBlock / 64];
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const bool v = i < n && gidx[i] != kNoMatch;
  const unsigned long long m = __ballot(v);
  if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}
__global__ __launch_bounds__(kBlock) void k_count_valid(const uint32_t* __restrict__ gidx, uint32_t n,
                                                        uint32_t* __restrict__ block_counts) {
  k_count_valid_body(gidx, n, block_counts);
}

__device__ __forceinline__ void k_scan_blocks_body(const uint32_t* __restrict__ counts, uint32_t nb,
                                                   uint32_t* __restrict__ offsets, uint32_t* __restrict__ total) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t base = 0; base < nb; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t c = i < nb ? counts[i] : 0;
    uint32_t incl = c;  // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up((int)incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t wpre = 0;
    for (int w = 0; w < wave; w++) wpre += wsum[w];
    if (i < nb) offsets[i] = carry + wpre + incl - c;
    __syncthreads();
    if (threadIdx.x == 1023) carry += wpre + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total) *total = carry;
}
__global__ __launch_bounds__(1024) void k_scan_blocks(const uint32_t* __restrict__ counts, uint32_t nb,
                                                      uint32_t* __restrict__ offsets, uint32_t* __restrict__ total) {
  k_scan_blocks_body(counts, nb, offsets, total);
}

__device__ __forceinline__ void k_compact_body(const uint32_t* __restrict__ gidx, const float4* __restrict__ pq,
                                               uint32_t n, const uint32_t* __restrict__ block_offsets,
                                               uint32_t* __restrict__ o_li, uint32_t* __restrict__ o_gi,
                                               float* __restrict__ o_x, float* __restrict__ o_y,
                                               float* __restrict__ o_z, float* __restrict__ o_d2) {
  __shared__ uint32_t wc[kBlock / 64];
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const uint32_t gi = i < n ? gidx[i] : kNoMatch;
  const bool v = gi != kNoMatch;
  const unsigned long long m = __ballot(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wc[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (!v) return;
  uint32_t pos = block_offsets[blockIdx.x] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; w++) pos += wc[w];
  const float4 q = pq[i];
  if (o_li) o_li[pos] = i;
  if (o_gi) o_gi[pos] = gi;
  if (o_x) o_x[pos] = q.x;
  if (o_y) o_y[pos] = q.y;
  if (o_z) o_z[pos] = q.z;
  if (o_d2) o_d2[pos] = q.w;
}
__global__ __launch_bounds__(kBlock) void k_compact(const uint32_t* __restrict__ gidx, const float4* __restrict__ pq,
                                                    uint32_t n, const uint32_t* __restrict__ block_offsets,
                                                    uint32_t* __restrict__ o_li, uint32_t* __restrict__ o_gi,
                                                    float* __restrict__ o_x, float* __restrict__ o_y,
                                                    float* __restrict__ o_z, float* __restrict__ o_d2) {
  k_compact_body(gidx, pq, n, block_offsets, o_li, o_gi, o_x, o_y, o_z, o_d2);
}
// the same three steps for every job of a batch (blockIdx.y = job), once the job's loop has terminated: the final
// pairings of job j land in its part of the batch's pairs block, ascending local index
__global__ __launch_bounds__(kBlock) void k_count_valid_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  if (blockIdx.x >= j.nb || !j.cp_out || !j.st->done) return;
  k_count_valid_body(j.pair_gidx, j.n, j.cp_counts);
}
__global__ __launch_bounds__(1024) void k_scan_blocks_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  if (!j.cp_out || !j.st->done) return;
  k_scan_blocks_body(j.cp_counts, j.nb, j.cp_counts + j.nb, nullptr);
}
__global__ __launch_bounds__(kBlock) void k_compact_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  if (blockIdx.x >= j.nb || !j.cp_out || !j.st->done) return;
  const uint32_t S = j.cp_stride;
  k_compact_body(j.pair_gidx, j.pair_q, j.n, j.cp_counts + j.nb, j.cp_out, j.cp_out + S, (float*)(j.cp_out + 2 * S),
                 (float*)(j.cp_out + 3 * S), (float*)(j.cp_out + 4 * S), (float*)(j.cp_out + 5 * S));
}

// point-to-plane pairings: flags + compaction in ascending local index
__global__ void k_pl_flags(const float4* __restrict__ pl_c, uint32_t n, uint32_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = pl_c[i].w != 0.f ? i : kNoMatch;
}
__global__ __launch_bounds__(kBlock) void k_compact_pl(const uint32_t* __restrict__ flags, const float4* __restrict__ pl_c,
                                                       const float4* __restrict__ pl_n, uint32_t n,
                                                       const uint32_t* __restrict__ block_offsets, uint32_t* __restrict__ o_li,
                                                       float* __restrict__ o_cx, float* __restrict__ o_cy,
                                                       float* __restrict__ o_cz, float* __restrict__ o_nx,
                                                       float* __restrict__ o_ny, float* __restrict__ o_nz) {
  __shared__ uint32_t wc[kBlock / 64];
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const bool v = i < n && flags[i] != kNoMatch;
  const unsigned long long m = __ballot(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wc[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (!v) return;
  uint32_t pos = block_offsets[blockIdx.x] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; w++) pos += wc[w];
  const float4 c = pl_c[i], nn = pl_n[i];
  if (o_li) o_li[pos] = i;
  if (o_cx) o_cx[pos] = c.x;
  if (o_cy) o_cy[pos] = c.y;
  if (o_cz) o_cz[pos] = c.z;
  if (o_nx) o_nx[pos] = nn.x;
  if (o_ny) o_ny[pos] = nn.y;
  if (o_nz) o_nz[pos] = nn.z;
}

// dense outputs of the un-compacted search
__global__ void k_unpack_dense(const uint32_t* __restrict__ gidx, const float4* __restrict__ pq, uint32_t n,
                               uint32_t* __restrict__ o_gi, float* __restrict__ o_x, float* __restrict__ o_y,
                               float* __restrict__ o_z, float* __restrict__ o_d2) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 q = pq[i];
  if (o_gi) o_gi[i] = gidx[i];
  if (o_x) o_x[i] = q.x;
  if (o_y) o_y[i] = q.y;
  if (o_z) o_z[i] = q.z;
  if (o_d2) o_d2[i] = q.w;
}

// solver-granular path: pack caller pairings into the pair buffers
__global__ void k_pack_pairs(const float* __restrict__ g3, uint32_t n, uint32_t stride, float4* __restrict__ pq,
                             uint32_t* __restrict__ gidx) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  pq[i] = make_float4(g3[i], g3[stride + i], g3[2 * stride + i], 0.f);
  gidx[i] = i;
}
