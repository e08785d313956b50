// mh_loop_wave.h -- k_icpw: the small layer's whole ICP loop in one launch with the plan / scan search (round 6).
//
// k_icp16 (round 5) gives every scan point a DPP row of 16 lanes: a 1.6 k-point layer -- the default pipeline's ICP layer,
// lidar3d-default.yaml:196-204 on the `decimated_for_icp` layer -- holds ~400 waves in ~52 workgroups of 512 lanes x ~250
// registers, one workgroup per CU.  Four such loops fill the part, so the alignments of more than four sequences took turns or
// went through lock-step batches that last as long as their slowest job (VERDICT r5 "what's weak" 4).
// Here the search is mh_nn_flat.h's: lane = point -> lane = candidate voxel -> lane = record, handed on through the wave's
// 7 KB of LDS (flat_search_points).  A wave owns kLwPtsPerWave points, a workgroup of kLwWaves waves kLwPoints = 128 of them:
// the same layer is ~50 waves in 13 workgroups of 256 lanes, two or more of which share a CU with each other and with other
// kernels, and the loops of sixteen sequences run side by side without anybody waiting for anybody.
//
// Everything else is k_icp16's, bit for bit: one column of 18 sums per GROUP of 32 points (a workgroup writes four), the sums
// of a group added in point order, every sum a 16-byte entry {value, serial, check word} in one agent-scope store, every
// workgroup fetching all columns of the step, adding them in rows_finish' order and closing the Gauss-Newton step on its own
// copy of the state block with solve_body.  The pairings are exact (the lexicographic minimum of (d2, scan position) over the
// 27-voxel block: tests/test_gpu_parity.py::test_wave_loop_*), so the trajectory file is the one k_icp16 and the k_step16 chain
// write.  Point-to-plane pairings (NDT maps) stay with k_icp16<true>.
// A workgroup that waits longer than kLoopDeadlineTicks (wall clock, not spin counts: ADVICE r5) for an entry gives up; the host
// then runs the alignment again launch by launch, as for k_icp16.
#pragma once

#ifndef MH_LW_WAVES
#define MH_LW_WAVES 4
#endif
#ifndef MH_LW_PTS
#define MH_LW_PTS 32
#endif
#ifndef MH_LW_MIN_WGS   // workgroups per CU the register allocator has to make room for (2: 256 registers per lane)
#define MH_LW_MIN_WGS 2
#endif
constexpr uint32_t kLwWaves = MH_LW_WAVES;
constexpr uint32_t kLwPtsPerWave = MH_LW_PTS;
constexpr uint32_t kLwThreads = kLwWaves * 64u;
constexpr uint32_t kLwPoints = kLwWaves * kLwPtsPerWave;      // points per workgroup
constexpr uint32_t kLwGroups = kLwPoints / kStepPoints;       // columns of sums per workgroup
static_assert(kLwPoints % kStepPoints == 0 && kLwPtsPerWave * 27 <= (uint32_t)FlatWaveSmall::kCands, "whole groups per workgroup; every candidate of a wave's points has a place in its list");

// rows_issue / loop_rows_fetch / rows_finish for a workgroup of NT lanes: the NVALS x kG (row, g) work items of the 512-lane
// kernels are dealt to the NT lanes round robin; the additions and their order are rows_finish' exactly.
template <int NVALS, int NT>
struct LwRows {
  typedef RowLoads<NVALS, (int)kLwMaxGroups> RL;
  static constexpr int kItems = NVALS * RL::kG;
  static constexpr int kV = (kItems + NT - 1) / NT;
  double v[kV][RL::kL];
};
template <int NVALS, int NT>
__device__ __forceinline__ void lw_rows_fetch(LwRows<NVALS, NT>& r, const AgentBuf& x, uint32_t base, uint32_t n, uint32_t serial,
                                              unsigned long long deadline, uint32_t* gave_up) {
  typedef typename LwRows<NVALS, NT>::RL RL;
  constexpr int kV = LwRows<NVALS, NT>::kV;
  uint32_t need = 0;
  uint32_t e0[kV];
#pragma unroll
  for (int u = 0; u < kV; u++) {
    const int vt = (int)threadIdx.x + u * NT;
    const int row = vt / RL::kG, g = vt % RL::kG;
    e0[u] = base + (uint32_t)(row < NVALS ? row : 0) * kLoopRowStride + (uint32_t)g;
#pragma unroll
    for (int j = 0; j < RL::kL; j++) {
      r.v[u][j] = 0.0;
      if (row < NVALS && (uint32_t)(g + j * RL::kG) < n) need |= 1u << (u * RL::kL + j);
    }
  }
  // Waiting costs the OTHER kernels on the device, not this one: an agent-scope load goes past the XCD's L2, and the workgroups
  // of sixteen loops re-reading every entry they miss keep the fabric busy (rocprofv3 trace of 16 sequences: the map updates'
  // and filters' small kernels 5 x slower beside them).  So a wave first polls ONE entry per column -- row `wave`, lane = column --
  // and fetches the step's entries only when those carry the serial number (the entries of a column leave its workgroup in one
  // store instruction; whatever has not landed yet is retried below, as before).
  {
    static_assert(kLwMaxGroups <= 128, "two columns per lane");
    const uint32_t lane = threadIdx.x & 63u, prow = (threadIdx.x >> 6) % (uint32_t)NVALS;
    const bool poll = lane < n, poll2 = lane + 64u < n;
    for (uint32_t spins = 0;; spins++) {
      const u32x4v w = __builtin_amdgcn_raw_buffer_load_b128(x.rsrc, (int)((base + prow * kLoopRowStride + (poll ? lane : 0u)) * 16u), 0, /*aux: sc1*/ 16);
      u32x4v w2 = w;
      if (n > 64u) w2 = __builtin_amdgcn_raw_buffer_load_b128(x.rsrc, (int)((base + prow * kLoopRowStride + (poll2 ? lane + 64u : 0u)) * 16u), 0, /*aux: sc1*/ 16);
      const bool here = (!poll || (w.z == serial && w.w == (w.x ^ w.y ^ w.z))) && (!poll2 || (w2.z == serial && w2.w == (w2.x ^ w2.y ^ w2.z)));
      if (__ballot(!here) == 0ull) break;
      if (((spins & 15u) == 15u && wall_clock64() > deadline) ||
          __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) {
        __hip_atomic_store(gave_up, 1u + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        need = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  for (uint32_t spins = 0; need; spins++) {
    u32x4v w[kV][RL::kL];
#pragma unroll
    for (int u = 0; u < kV; u++)
#pragma unroll
      for (int j = 0; j < RL::kL; j++)
        if ((need >> (u * RL::kL + j)) & 1u)
          w[u][j] = __builtin_amdgcn_raw_buffer_load_b128(x.rsrc, (int)((e0[u] + (uint32_t)(j * RL::kG)) * 16u), 0, /*aux: sc1*/ 16);
#pragma unroll
    for (int u = 0; u < kV; u++)
#pragma unroll
      for (int j = 0; j < RL::kL; j++)
        if (((need >> (u * RL::kL + j)) & 1u) && w[u][j].z == serial && w[u][j].w == (w[u][j].x ^ w[u][j].y ^ w[u][j].z)) {
          r.v[u][j] = __hiloint2double((int)w[u][j].y, (int)w[u][j].x);
          need &= ~(1u << (u * RL::kL + j));
        }
    if (!need) break;
    if (((spins & 15u) == 15u && wall_clock64() > deadline) ||
        __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) {  // give up loudly
      __hip_atomic_store(gave_up, 1u + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}
template <int NVALS, int NT>
__device__ __forceinline__ void lw_rows_finish(const LwRows<NVALS, NT>& r, uint32_t n, double* __restrict__ out, double (*red)[64]) {
  typedef typename LwRows<NVALS, NT>::RL RL;
  constexpr int G = RL::kG;
  constexpr int kV = LwRows<NVALS, NT>::kV;
#pragma unroll
  for (int u = 0; u < kV; u++) {
    const int vt = (int)threadIdx.x + u * NT;
    const int row = vt / G, g = vt % G;
    if (row < NVALS) {
      const uint32_t full = (n > (uint32_t)(g + 7 * G)) ? 1u + (n - (uint32_t)(g + 7 * G) - 1u) / (8u * G) : 0u;  // rows_finish' rounds of eight
      double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int j = 0; j < RL::kL; j++) {
        if ((uint32_t)j < 8u * full) s[j % 8] += r.v[u][j];
        else if ((uint32_t)(g + j * G) < n) s[0] += r.v[u][j];
      }
      red[row][g] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < NVALS) {
    double part[G];
#pragma unroll
    for (int q = 0; q < G; q++) part[q] = red[threadIdx.x][q];
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < G; q++) acc += part[q];
    out[threadIdx.x] = acc;
  }
  __syncthreads();
}

struct LwShared {
  SolveSharedTotals sh;
  __attribute__((aligned(8))) uint32_t lst_raw[kStateHeadDwords];
  // the points' sums of a body (written after the step is closed, read by the group sums) AND the scratch of the ordered sums over
  // the columns (lw_rows_finish, before the step is closed): never live together, workgroup barriers in between
  union {
    double rowsA[kAccN][kLwPoints + 1];
    double red[kAccN][64];
  } u;
  FlatWaveSmall fw[kLwWaves];
  uint32_t gave_up;
};

// the loop of ONE alignment: workgroup `wg` of `nwg`, exchange block `xa`, entries numbered from serial0
__device__ __forceinline__ void icpw_body(LwShared& S, IcpDeviceState* s_canon, const MatchK* __restrict__ kp, const SolveK* __restrict__ sk,
                                          const float* __restrict__ lx, const float* __restrict__ ly, const float* __restrict__ lz,
                                          uint32_t n, const MapView& map, float4* pair_q, uint32_t* pair_gidx, void* xa, uint32_t ngroups,
                                          uint32_t wg, uint32_t serial0, uint32_t max_steps, uint32_t want_cov) {
  const uint32_t tid = threadIdx.x;
  IcpDeviceState* const lst = reinterpret_cast<IcpDeviceState*>(S.lst_raw);
  const AgentBuf bxa = agent_buf(xa, 2u * kAccN * kLoopRowStride);
  const uint32_t wave = tid >> 6, lane = tid & 63u;
  const bool mine = lane < kLwPtsPerWave;                       // this lane holds a point of the wave
  const uint32_t slot = wave * kLwPtsPerWave + (mine ? lane : 0u);  // its place in the workgroup's 128 rows
  const uint32_t i = wg * kLwPoints + slot, ic = i < n ? i : n - 1;
  const bool in = mine && i < n;
  const float x = G(lx)[ic], y = G(ly)[ic], z = G(lz)[ic];
  if (tid < kStateHeadDwords) S.lst_raw[tid] = G(reinterpret_cast<const uint32_t*>(s_canon))[tid];  // (uploaded before the launch)
  if (tid == 0) S.gave_up = 0;
  __syncthreads();
  typedef const MatchK __attribute__((address_space(4))) * cmatchk_ptr;
  const cmatchk_ptr ck = (cmatchk_ptr)uniform_const_ptr(kp);
  const uint32_t kernel = ck->kernel;
  const unsigned long long deadline = wall_clock64() + kLoopDeadlineTicks;
  // the pairing of this lane's point: found at an iteration's start, used by its inner steps and as the next search's bound
  f32x4 q = (f32x4){0.f, 0.f, 0.f, __builtin_inff()};
  uint32_t gidx = kNoMatch;
  bool ok = false, searched = false;
  uint32_t step = 0;
  MH_LOOP_STAMPS;
#pragma nounroll
  for (;; step++) {
    MH_LOOP_STAMP(0);
    if (lst->pending) {  // the sums of step - 1 (serial0 + step), every workgroup for itself
      const uint32_t half = (step - 1u) & 1u;
      LwRows<kAccN, (int)kLwThreads> ra;
      lw_rows_fetch<kAccN, (int)kLwThreads>(ra, bxa, half * kAccN * kLoopRowStride, ngroups, serial0 + step, deadline, &S.gave_up);
      MH_LOOP_STAMP(1);
      lw_rows_finish<kAccN, (int)kLwThreads>(ra, ngroups, S.sh.totA, S.u.red);
      MH_LOOP_STAMP(2);
      if (S.gave_up) break;  // (behind lw_rows_finish' barriers: the same in every wave)
      solve_body<true, false>(lst, sk, nullptr, 0u, 0u, nullptr, 0u, 0u, S.sh, true, false);
      if (tid == 0) lst->pending = 0u;
      __syncthreads();
      MH_LOOP_STAMP(3);
    }
    if (lst->done || step >= max_steps) break;
    const uint32_t inner = lst->inner, iter = lst->iter;
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = lst->T[k];
    const float thr2 = lst->cur_thr2, ang2 = lst->cur_ang2;
    const double kparam = lst->cur_kparam;
    Acc a;
    acc_zero(a);
    if (inner == 0) {  // (workgroup-uniform) a new ICP iteration: the matcher
      float px, py, pz;
      transform_point(T, x, y, z, px, py, pz);
      float bound0 = __builtin_inff();
      if (iter > 0 && !map.no_prev_bound && q.w < __builtin_inff()) {
        const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
        bound0 = (dx * dx + dy * dy) + dz * dz;  // the candidate arithmetic of the scans
      }
      MH_LOOP_STAMP(6);
      const FlatHit r = flat_search_points(S.fw[wave], map, in, px, py, pz, bound0);
      MH_LOOP_STAMP(7);
      const float n2 = (px * px + py * py) + pz * pz;
      ok = r.found && (r.d2 < thr2 + ang2 * n2);
      q = (f32x4){r.pt.x, r.pt.y, r.pt.z, r.d2};
      gidx = ok ? __float_as_uint(r.pt.w) : kNoMatch;
      searched = true;
    }
    if (in) acc_pt2pt_masked(a, T, ok, x, y, z, q.x, q.y, q.z, kernel, kparam, ck->w_pt2pt);
    if (mine) {
#pragma unroll
      for (int j = 0; j < kAccN; j++) S.u.rowsA[j][slot] = a.v[j];
    }
    MH_LOOP_STAMP(8);
    __syncthreads();
    MH_LOOP_STAMP(4);
    const uint32_t out = step & 1u;
    if (tid < kAccN * kLwGroups) {  // one lane per (group, sum): the group's 32 rows in point order
      const uint32_t grp = tid / kAccN, j = tid % kAccN, g = wg * kLwGroups + grp;
      if (g < ngroups) {
        const double* rowp = &S.u.rowsA[j][grp * kStepPoints];
        double sum = rowp[0];
#pragma unroll
        for (int r = 1; r < (int)kStepPoints; r++) sum += rowp[r];
        loop_entry_store(bxa, (out * kAccN + j) * kLoopRowStride + g, sum, serial0 + step + 1u);
      }
    }
    if (tid == 0) lst->pending = 1u;
    __syncthreads();
    MH_LOOP_STAMP(5);
  }
  MH_LOOP_STAMPS_OUT(step);
  if (S.gave_up) {  // the canonical block keeps done == 0: the host runs the alignment again, launch by launch
    if (tid == 0) {
      atomicAdd(&s_canon->handover_timeouts, 1u);
      if (atomicCAS(&s_canon->dbg[0], 0u, 5u) == 0u) {
        s_canon->dbg[1] = wg; s_canon->dbg[2] = S.gave_up - 1u; s_canon->dbg[3] = serial0 + step; s_canon->dbg[4] = step; s_canon->dbg[5] = ngroups;
      }
    }
    return;
  }
  // what the covariance kernels and the pairing export read once the loop has ended: the pairings of the last search
  if (in && searched) {
    pair_q[i] = make_float4(q.x, q.y, q.z, q.w);
    G(pair_gidx)[i] = gidx;
  }
  if (wg == 0) {
    if (tid < kStateHeadDwords) G(reinterpret_cast<uint32_t*>(s_canon))[tid] = S.lst_raw[tid];
    if (want_cov && lst->done && tid < 6) {  // k_cov_prepare's six lanes: the covariance chain that follows starts at k_cov_accum
      Pose Tc;
#pragma unroll
      for (int k = 0; k < 12; k++) Tc.m[k] = lst->T[k];
      double outv[12];
      cov_prepare_lane(Tc, (int)tid, sk->cov_hx, sk->cov_ha, outv);
#pragma unroll
      for (int k = 0; k < 12; k++) s_canon->covD[tid * 12 + k] = outv[k];
    }
  }
}

__global__ __launch_bounds__(kLwThreads, MH_LW_MIN_WGS) void k_icpw(IcpDeviceState* s_canon, const MatchK* __restrict__ kp, const SolveK* __restrict__ sk,
                                                     const float* __restrict__ lx, const float* __restrict__ ly,
                                                     const float* __restrict__ lz, uint32_t n, MapView map, float4* pair_q,
                                                     uint32_t* pair_gidx, void* xa, uint32_t ngroups, uint32_t serial0,
                                                     uint32_t max_steps, uint32_t want_cov) {
  __shared__ LwShared S;
  if (blockIdx.x * kLwGroups >= ngroups) return;
  icpw_body(S, s_canon, kp, sk, lx, ly, lz, n, map, pair_q, pair_gidx, xa, ngroups, blockIdx.x, serial0, max_steps, want_cov);
}

// k_icpw_b: the same loop for the jobs of a lock-step group, side by side in ONE launch (blockIdx.y = job): every job has its own
// workgroups (a quarter of k_icp16_b's per point, so sixteen 1.6 k-point jobs are 208 workgroups, two to a CU, and nobody takes
// several groups), exchanges among its own, and ends when it ends.  The covariance chain of the batch follows as for k_icp16_b.
__global__ __launch_bounds__(kLwThreads, MH_LW_MIN_WGS) void k_icpw_b(const BatchJob* __restrict__ jobs) {
  __shared__ LwShared S;
  const BatchJob& j = jobs[blockIdx.y];
  const uint32_t n = j.n;
  const uint32_t ngroups = (n + kStepPoints - 1) / kStepPoints;
  if (n == 0 || blockIdx.x * kLwGroups >= ngroups) return;
  const SolveK* const sk = j.sk;
  const uint32_t max_steps = j.loop_pad ? 1u : sk->max_iterations * sk->max_inner + 1u;  // (loop_pad: MH_LOOP16_TEST_ABANDON)
  icpw_body(S, j.st, j.mk, sk, j.lx, j.ly, j.lz, n, j.map, j.pair_q, j.pair_gidx, j.loop_xa, ngroups, blockIdx.x, j.loop_serial0, max_steps, 0u);
}
