// mh_k_launch.h -- the __global__ entry points of the kernel bodies: one alignment per launch, or one job per blockIdx.y (*_b, the
// lock-step batches of mh_icp_align_batch).
#pragma once

// ---- kernel entry points of the bodies above: one alignment per launch, or one job per blockIdx.y -------------------
__global__ __launch_bounds__(kBlock, MH_QUAD_WAVES) void k_match4(const IcpDeviceState* __restrict__ st,
                                                   const float* __restrict__ lx, const float* __restrict__ ly,
                                                   const float* __restrict__ lz, uint32_t n, MapView map,
                                                   float4* __restrict__ pair_q, uint32_t* __restrict__ pair_gidx,
                                                   const uint32_t* __restrict__ perm
#ifdef MH_DEBUG_WAVETRACE
                                                   , unsigned long long* __restrict__ wtrace
#endif
) {
  k_match4_body(st, lx, ly, lz, n, map, pair_q, pair_gidx, perm
#ifdef MH_DEBUG_WAVETRACE
                , wtrace
#endif
  );
}
// ================================================================================================
// k_match_flat: the plan / scan matcher (mh_nn_flat.h) -- a wave per 64 consecutive scan points, per-point, per-voxel and
// per-record work each spread over all 64 lanes and handed on through the wave's slice of LDS.  Device-state driven like
// k_match4 (same arguments, same pairings, bit for bit); the first Gauss-Newton accumulation is the k_accum launch that follows.
// ================================================================================================
#ifndef MH_FLAT_WAVES
#define MH_FLAT_WAVES 6  // waves per SIMD the register allocator has to leave room for (80 VGPRs); the LDS allows 5.5
#endif
#ifndef MH_FLAT_THREADS
#define MH_FLAT_THREADS 64
#endif
constexpr uint32_t kFlatThreads = MH_FLAT_THREADS;  // ONE wave per workgroup (nothing is shared between waves): 0.1637 ms per launch against 0.1787 with two and 0.1838 with four -- a workgroup holds its slot until its slowest wave is done
constexpr uint32_t kFlatPointsPerBlock = kFlatThreads; // a lane per point in phase A
__device__ __forceinline__ uint32_t nblk_flat_dev(uint32_t n) { return (n + kFlatPointsPerBlock - 1u) / kFlatPointsPerBlock; }
// (the grid width is a multiple of 8 = the XCDs a launch is dealt over, whatever the layer's size)
constexpr uint32_t kFlatGridUnit = 8u;
inline uint32_t nblk_flat(size_t n) { return (uint32_t)(((n + kFlatPointsPerBlock - 1) / kFlatPointsPerBlock + kFlatGridUnit - 1) / kFlatGridUnit * kFlatGridUnit); }
__device__ __forceinline__ void k_match_flat_body(const IcpDeviceState* __restrict__ st, const float* __restrict__ lx,
                                                  const float* __restrict__ ly, const float* __restrict__ lz, uint32_t n,
                                                  MapView map, float4* __restrict__ pair_q, uint32_t* __restrict__ pair_gidx,
                                                  const uint32_t* __restrict__ perm, uint32_t block_x) {
  __shared__ FlatWave sh[kFlatThreads / 64];
  typedef const IcpDeviceState __attribute__((address_space(4))) * cstate_ptr;
  const cstate_ptr cst = (cstate_ptr)uniform_const_ptr(st);
  if (cst->done) return;  // grid-uniform
  const uint32_t bx = block_x;
  const uint32_t i0 = bx * kFlatPointsPerBlock + (threadIdx.x & ~63u);
  if (i0 >= n) return;    // whole waves
  const bool have_prev = cst->iter > 0 && !map.no_prev_bound;
  double T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = cst->T[k];
  match_flat_wave(sh[threadIdx.x >> 6], map, T, cst->cur_thr2, cst->cur_ang2, have_prev, lx, ly, lz, n, i0, pair_q, pair_gidx, perm);
}
__global__ __launch_bounds__(kFlatThreads, MH_FLAT_WAVES) void k_match_flat(const IcpDeviceState* __restrict__ st, const float* __restrict__ lx,
                                                             const float* __restrict__ ly, const float* __restrict__ lz, uint32_t n,
                                                             MapView map, float4* __restrict__ pair_q,
                                                             uint32_t* __restrict__ pair_gidx, const uint32_t* __restrict__ perm) {
  k_match_flat_body(st, lx, ly, lz, n, map, pair_q, pair_gidx, perm, blockIdx.x);
}
__global__ __launch_bounds__(kBlock, MH_QUAD_WAVES) void k_match4_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  k_match4_body(j.st, j.lx, j.ly, j.lz, j.n, j.map, j.pair_q, j.pair_gidx, nullptr
#ifdef MH_DEBUG_WAVETRACE
                , nullptr
#endif
  );
}
// A whole job per XCD.  Workgroup L of a launch runs on XCD L % 8 (tools/xcd_exchange.hip: 0 exceptions in 256; the grid's width is
// a multiple of 8), so with blockIdx.y = job every job's workgroups are dealt over all eight XCDs and every XCD's L2 fetches its own
// copy of every job's map: (2 FETCH_SIZE + WRITE_SIZE) = 1.64 x the compulsory bytes on C2.  Here XCD c takes the jobs c, c + 8, ...
// of the first 8 * floor(jobs / 8) one after the other (the rest keep the plain order): a map is fetched into ONE L2 --
// FETCH_SIZE 155.5 -> 89.7 MB per launch of 32 scans, traffic 1.64 -> 1.08 x compulsory, L2 hit rate 47 -> 69 %, headline
// 6320-6330 -> 6470-6480 scans/s.  (Unlike a contiguous PART of one scan per XCD -- profiles/r05_match_kernel.md section 5: 1.22 x
// at -37 % speed -- whole scans are equal work.)  -DMH_FLAT_NO_JOB_XCD: the plain order (A/B).
__global__ __launch_bounds__(kFlatThreads, MH_FLAT_WAVES) void k_match_flat_b(const BatchJob* __restrict__ jobs) {
  uint32_t job = blockIdx.y, bx = blockIdx.x;
  const uint32_t whole = gridDim.y & ~7u;  // jobs that are dealt an XCD each
  if (blockIdx.y < whole) {
    const uint32_t L = blockIdx.x + blockIdx.y * gridDim.x, xcd = L % 8u, slot = L / 8u;
    job = xcd + 8u * (slot / gridDim.x);
    bx = slot % gridDim.x;
  }
  const BatchJob& j = jobs[job];
  k_match_flat_body(j.st, j.lx, j.ly, j.lz, j.n, j.map, j.pair_q, j.pair_gidx, nullptr, bx);
}
#ifdef MH_DEV_VARIANTS
#include "mh_dev_variants.h"  // k_match_tile*, k_match_wave_*, k_match4o_b: development library only
#endif
template <bool SIGNED>
__global__ __launch_bounds__(kBlock, MH_ACCUM_WAVES) void k_accum(const IcpDeviceState* __restrict__ st, uint32_t first,
                                                  const MatchK* __restrict__ kp, const float* __restrict__ lx,
                                                  const float* __restrict__ ly, const float* __restrict__ lz, uint32_t n,
                                                  const float4* __restrict__ pair_q,
                                                  const uint32_t* __restrict__ pair_gidx, double* __restrict__ partials,
                                                  uint32_t pstride) {
  k_accum_body<SIGNED>(st, first, kp, lx, ly, lz, n, pair_q, pair_gidx, partials, pstride, blockIdx.x);
}
template <bool SIGNED>
__global__ __launch_bounds__(kBlock, MH_ACCUM_WAVES) void k_accum_b(const BatchJob* __restrict__ jobs, uint32_t first) {
  // (the matcher's job -> XCD mapping was tried here as well -- the pairings this launch reads were written through that XCD's L2 --
  //  and changes nothing: 22.4-22.9 us either way; a launch boundary leaves nothing of them in the L2)
  const BatchJob& j = jobs[blockIdx.y];
  const uint32_t bx = blockIdx.x;
  if (bx >= j.nba) return;
  k_accum_body<SIGNED>(j.st, first, j.mk, j.lx, j.ly, j.lz, j.n, j.pair_q, j.pair_gidx, j.part, j.nba, bx);
}
__global__ __launch_bounds__(kSolveThreads) void k_solve(IcpDeviceState* __restrict__ st, const SolveK* __restrict__ kp,
                                                         const double* __restrict__ partA, uint32_t nA, uint32_t strideA,
                                                         const double* __restrict__ partB, uint32_t nB,
                                                         uint32_t strideB, uint32_t first) {
  k_solve_body(st, kp, partA, nA, strideA, partB, nB, strideB, first);
}
__global__ __launch_bounds__(kSolveThreads) void k_solve_b(const BatchJob* __restrict__ jobs, uint32_t first) {
  const BatchJob& j = jobs[blockIdx.y];
  const uint32_t cols = first ? j.nbm : j.nba;  // the first step's partials come from the matcher-side producer
  k_solve_body(j.st, j.sk, j.part, cols, cols, nullptr, 0u, 0u, first);
}
__global__ void k_cov_prepare(IcpDeviceState* __restrict__ st, const SolveK* __restrict__ kp, uint32_t force) {
  k_cov_prepare_body(st, kp, force);
}
__global__ void k_cov_prepare_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  k_cov_prepare_body(j.st, j.sk, 0u);
}
__global__ __launch_bounds__(kBlock) void k_cov_accum(const IcpDeviceState* __restrict__ st, uint32_t force,
                                                      const float* __restrict__ lx, const float* __restrict__ ly,
                                                      const float* __restrict__ lz, uint32_t n,
                                                      const uint32_t* __restrict__ pair_gidx,
                                                      double* __restrict__ partials, uint32_t pstride) {
  k_cov_accum_body(st, force, lx, ly, lz, n, pair_gidx, partials, pstride);
}
__global__ __launch_bounds__(kBlock) void k_cov_accum_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  if (blockIdx.x >= j.nb) return;
  k_cov_accum_body(j.st, 0u, j.lx, j.ly, j.lz, j.n, j.pair_gidx, j.part, j.nb);
}
__global__ __launch_bounds__(kSolveThreads) void k_cov_finalize(IcpDeviceState* __restrict__ st, uint32_t force,
                                                                const double* __restrict__ partA, uint32_t nA,
                                                                uint32_t strideA, const double* __restrict__ partB,
                                                                uint32_t nB, uint32_t strideB) {
  k_cov_finalize_body(st, force, partA, nA, strideA, partB, nB, strideB);
}
__global__ __launch_bounds__(kSolveThreads) void k_cov_finalize_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  k_cov_finalize_body(j.st, 0u, j.part, j.nb, j.nb, j.partb, j.partb ? j.nb : 0u, j.partb ? j.nb : 0u);
}
__global__ __launch_bounds__(kBlock) void k_cov_accum_plbuf_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  if (blockIdx.x >= j.nb || !j.partb) return;
  k_cov_accum_plbuf_body(j.st, j.lx, j.ly, j.lz, j.n, j.pl_c, j.pl_n, j.partb, j.nb);
}
template <bool PL, bool FUSED>
__global__ __launch_bounds__(kBlock) void k_match16(const IcpDeviceState* __restrict__ st, const MatchK* __restrict__ kp,
                                                    const float* __restrict__ lx, const float* __restrict__ ly,
                                                    const float* __restrict__ lz, uint32_t n, MapView map,
                                                    float4* __restrict__ pair_q, uint32_t* __restrict__ pair_gidx,
                                                    float4* __restrict__ pl_c, float4* __restrict__ pl_n,
                                                    double* __restrict__ partials, uint32_t pstride) {
  k_match16_body<PL, FUSED>(st, kp, lx, ly, lz, n, map, pair_q, pair_gidx, pl_c, pl_n, partials, pstride);
}
// row kernel with the fused first accumulation, one job per blockIdx.y (layers of 2-12 k points in lock step)
__global__ __launch_bounds__(kBlock) void k_match16f_b(const BatchJob* __restrict__ jobs) {
  const BatchJob& j = jobs[blockIdx.y];
  if (blockIdx.x >= j.nbm) return;
  k_match16_body<false, true>(j.st, j.mk, j.lx, j.ly, j.lz, j.n, j.map, j.pair_q, j.pair_gidx, nullptr, nullptr, j.part, j.nbm);
}
// start of a lock-step batch: the staged [state | params | schedules] of all jobs -> where each job keeps them
__global__ void k_scatter_blocks(const BatchJob* __restrict__ jobs, const uint32_t* __restrict__ stage, uint32_t dwords) {
  const BatchJob& j = jobs[blockIdx.x];
  uint32_t* dst = reinterpret_cast<uint32_t*>(j.st);
  const uint32_t* src = stage + j.stage_off;
  for (uint32_t i = threadIdx.x; i < dwords; i += blockDim.x) dst[i] = src[i];
  for (uint32_t i = threadIdx.x; i < j.sched_dwords; i += blockDim.x) j.sched_dst[i] = src[dwords + i];
}
// all jobs' state blocks into one contiguous buffer: one read-back per chunk instead of one per job
__global__ void k_gather_states(const BatchJob* __restrict__ jobs, IcpDeviceState* __restrict__ out) {
  const uint32_t* src = reinterpret_cast<const uint32_t*>(jobs[blockIdx.x].st);
  uint32_t* dst = reinterpret_cast<uint32_t*>(out + blockIdx.x);
  for (uint32_t i = threadIdx.x; i < sizeof(IcpDeviceState) / 4; i += blockDim.x) dst[i] = src[i];
}
