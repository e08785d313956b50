// mh_k_step.h -- the small layer's loop: k_step16 (search + sums per launch, the Gauss-Newton step carried into the next launch),
// k_icp16 / k_icp16_b (the whole loop in ONE launch, a DPP row per point) and, through mh_loop_wave.h, k_icpw / k_icpw_b (the same
// loop with the plan / scan search).
#pragma once

// ================================================================================================
// k_step16: the small-layer iteration with the solve CARRIED INTO THE NEXT LAUNCH (round 4).  The chain it replaces spent a
// launch of ONE workgroup on every Gauss-Newton step (k_match16 | k_accum_solve1 | k_accum_solve1, round 3: 28 us per ICP iteration of
// the real pipeline's 1.2-1.6 k-point layer).  Here every launch is the same kernel over the whole layer, and what it does
// is decided by the state block alone:
//   1. every workgroup copies the state block's head into LDS and, if a step is pending, closes it: the ordered sum of the
//      partials of the previous launch + solve_body -- all workgroups compute the same bits, nobody waits for a hand-over;
//      workgroup 0 writes the new state to ANOTHER state block (a ping-pong pair beside the canonical block: the one a launch
//      reads is never written by it) and publishes the progress word;
//   2. body: at the start of an ICP iteration (inner == 0) the row search of k_match16 for groups of 32 points, the pairings
//      stored and their Gauss-Newton sums written as one partial column per GROUP (also ping-pong: other workgroups may
//      still be reading the previous launch's); at an inner step the sums of the stored pairings under the new pose.
// An ICP iteration is max_inner launches (2 in the shipped pipelines) instead of 1 + max_inner, and a launch never idles
// because an iteration closed early: the next one simply starts in its place.  A workgroup takes the groups wg, wg + nw, ...:
// one group each for a single alignment; in a lock-step batch the host caps nw so that all jobs' workgroups are resident
// at once (512 threads x ~250 registers: one workgroup per CU) -- the columns, hence the sums and the result bit for bit, do
// not depend on nw.  `close_only` (one workgroup, in place into the canonical block): the pending step at the end of a
// chunk of launches.
// Phase stamps (tools/phase_probe.py, 1.4 k points, us): state into LDS 0.2, partials summed 1.4, assemble 0.4, LDLT 0.7,
// exp + compose 0.55, log 1.0, tail 0.6, state written 0.4, search + accumulate 1.1, sums 1.0 -- 7.5 of the ~13 us from one
// launch to the next; the rest is the launch.
// (Also built and measured in round 4, and removed: k_loop16, the whole loop in ONE launch -- the same body and solve per
// workgroup, a grid barrier between them (arrival counter + agent-scope loads of the partials; no cache invalidation, the map
// stays in the L2s).  Bit-identical, and slower: 15.8 us per Gauss-Newton step against 16.2 launch by launch in a 40-iteration
// fit, 0.594-0.611 against 0.544-0.551 ms of ICP per scan on the city drive (0.617 with an L2 write-back as the release).
// Crossing the XCDs costs what a launch boundary costs, and inside a loop the compiler hoists ~470 bytes per lane of lane
// masks, offset tables and literal constants into scratch.)
// (And: ONE launch per iteration, k_iter16 -- no sums cross workgroups at all: every workgroup accumulates ALL points of the layer
// for both Gauss-Newton steps of the iteration the previous launch matched, solves them on its own copy of the state, then
// searches its own groups.  Same trajectory file as k_step16's; 0.727-0.74 against 0.553 ms of ICP per scan: three rounds of
// agent-scope loads of the stored pairings per step and a called (not inlined: spills) solve cost more than the launch they
// save.  Removed.)
// (And: the covariance + the result written to the host's page-locked mirror by the launch that finds the loop finished --
// no covariance launches, no read-back copy, no event -- bit-identical to the three covariance kernels, 1536-1548 -> 1527-1573
// scans/s: the one workgroup that sums the whole layer takes what the launches took.  Removed.)
// ================================================================================================
constexpr uint32_t kStepPoints = kSolveThreads / 16;  // scan points (DPP rows) per group
constexpr uint32_t kStepRowsA = kAccN + 1;        // rows of one half of the point-to-point partials: the sums + the column's tag
constexpr uint32_t kStepRowsB = kGenN + 1;        // ... of the point-to-plane partials
constexpr uint32_t kStepMaxPoints = 8192;      // (above: k_match16<fused> | k_solve | k_accum | k_solve, then the quad matcher's chain)
constexpr uint32_t kStepMaxWorkgroups = 256;   // one per CU (512 threads x ~250 registers); beyond, workgroups take several groups
constexpr uint32_t kStateHeadDwords = (uint32_t)(offsetof(IcpDeviceState, cov) / 4);  // everything the loop touches
constexpr uint32_t kStateSerialDword = (uint32_t)(offsetof(IcpDeviceState, serial) / 4);
static_assert(kStateHeadDwords <= 64 && offsetof(IcpDeviceState, cov) % 8 == 0, "state head is copied by one wave");

// reduce_rows with agent-scope loads, all of a lane's loads issued before any sum, the sums in reduce_rows' order exactly (lane (row, g) adds columns g, g + G, ...: eight partial sums over the full rounds, the rest into
// the first, then the tree).  NVALS rows, up to kStepMaxPoints / kStepPoints columns.
template <int NVALS, int MAXCOLS = (int)(kStepMaxPoints / kStepPoints)>
struct RowLoads {
  static constexpr int kG = ((int)kSolveThreads / NVALS) > 64 ? 64 : ((int)kSolveThreads / NVALS);
  static constexpr int kL = (MAXCOLS + kG - 1) / kG;
  double v[kL];
};
template <int NVALS>
__device__ __forceinline__ void rows_issue(RowLoads<NVALS>& r, const double* part, uint32_t n, uint32_t stride) {
  constexpr int G = RowLoads<NVALS>::kG;
  const int row = (int)threadIdx.x / G, g = (int)threadIdx.x % G;
  const double* src = part + (size_t)(row < NVALS ? row : 0) * stride;
#pragma unroll
  for (int j = 0; j < RowLoads<NVALS>::kL; j++) {
    const uint32_t b = (uint32_t)(g + j * G);
    r.v[j] = (row < NVALS && b < n) ? __hip_atomic_load(src + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
  }
}
template <int NVALS, int MAXCOLS>
__device__ __forceinline__ void rows_finish(const RowLoads<NVALS, MAXCOLS>& r, uint32_t n, double* __restrict__ out, double (*red)[64]) {
  constexpr int G = RowLoads<NVALS, MAXCOLS>::kG;
  const int t = threadIdx.x, row = t / G, g = t % G;
  if (row < NVALS) {
    const uint32_t full = (n > (uint32_t)(g + 7 * G)) ? 1u + (n - (uint32_t)(g + 7 * G) - 1u) / (8u * G) : 0u;  // reduce_rows' rounds of eight
    double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < RowLoads<NVALS, MAXCOLS>::kL; j++) {
      if ((uint32_t)j < 8u * full) s[j % 8] += r.v[j];
      else if ((uint32_t)(g + j * G) < n) s[0] += r.v[j];
    }
    red[row][g] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  }
  __syncthreads();
  if (t < NVALS) {  // (all G reads first, then the additions in order: a read per addition costs its LDS latency G times over)
    double part[G];
#pragma unroll
    for (int q = 0; q < G; q++) part[q] = red[t][q];
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < G; q++) acc += part[q];
    out[t] = acc;
  }
  __syncthreads();
}

// Stored pairings cross launches of the k_step16 chain the way its state and partial sums do: agent-scope
// (write-through) stores, acknowledged before the group's partial column is tagged, and agent-scope loads by the launch that
// has seen the tag -- never answered from a stale L1 / L2 line, whichever XCD the reader runs on.
// (16 bytes in one sc1 access through a buffer descriptor -- an agent-scope __hip_atomic lowers to sc1 only up to 8 bytes, and
// 8-byte write-through stores cost 2.7x the 16-byte ones per byte -- with the compiler tracking the load like any other.)
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
struct AgentBuf {
  __amdgpu_buffer_rsrc_t rsrc;
};
__device__ __forceinline__ AgentBuf agent_buf(const void* base, uint32_t n_records) {
  const unsigned long long a = (unsigned long long)base;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  void* p = (void*)(((unsigned long long)hi << 32) | lo);
  AgentBuf b;
  b.rsrc = __builtin_amdgcn_make_buffer_rsrc(p, (short)0, (int)__builtin_amdgcn_readfirstlane((int)(n_records * 16u)), 0x00020000);
  return b;
}
__device__ __forceinline__ void store_agent_b128(const AgentBuf& b, uint32_t i, f32x4 v) {
  const u32x4v w = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(w, b.rsrc, (int)(i * 16u), 0, /*aux: sc1*/ 16);
}
__device__ __forceinline__ f32x4 load_agent_b128(const AgentBuf& b, uint32_t i) {
  const u32x4v w = __builtin_amdgcn_raw_buffer_load_b128(b.rsrc, (int)(i * 16u), 0, /*aux: sc1*/ 16);
  return (f32x4){__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w)};
}

template <bool PL>
__device__ __forceinline__ void k_step16_body(const IcpDeviceState* __restrict__ s_in, IcpDeviceState* s_out,
                                              IcpDeviceState* s_canon, const MatchK* __restrict__ kp,
                                              const SolveK* __restrict__ sk, const float* __restrict__ lx,
                                              const float* __restrict__ ly, const float* __restrict__ lz, uint32_t n,
                                              MapView map, float4* pair_q, uint32_t* pair_gidx, float4* pl_c, float4* pl_n,
                                              const double* __restrict__ partA_in, double* __restrict__ partA_out,
                                              const double* __restrict__ partB_in, double* __restrict__ partB_out,
                                              uint32_t ngroups, uint32_t nw, uint32_t close_only, uint32_t expect, uint32_t expect_rel) {
  __shared__ SolveShared sh;
  __shared__ __attribute__((aligned(8))) uint32_t lst_raw[kStateHeadDwords];
  __shared__ double rowsA[kAccN][kStepPoints + 1];
  __shared__ double rowsB[PL ? kGenN : 1][kStepPoints + 1];
  const AgentBuf b_pair = agent_buf(pair_q, n), b_plc = agent_buf(PL ? (const void*)pl_c : (const void*)pair_q, n),
                 b_pln = agent_buf(PL ? (const void*)pl_n : (const void*)pair_q, n);
  __shared__ uint32_t pair_acks;  // waves whose stores of the current and earlier groups are acknowledged
  uint32_t acks_wanted = 0;
  const uint32_t tid = threadIdx.x, wg = blockIdx.x;
  if (wg >= nw) return;
  if (tid == 0) pair_acks = 0;  // (barriers below before anybody counts)  // (lock-step batches: the grid is the largest job's)
  IcpDeviceState* const lst = reinterpret_cast<IcpDeviceState*>(lst_raw);
  // the point of this row is on its way before the state is looked at.  (Not so what the PREVIOUS launch stored for it: see below.)
  const uint32_t row = tid >> 4, r16 = tid & 15u;
  uint32_t g = wg;
  uint32_t i = g * kStepPoints + row;
  uint32_t ic = i < n ? i : n - 1;
  float x = G(lx)[ic], y = G(ly)[ic], z = G(lz)[ic];
  // (and the tag of "its" column of the partials: checked against the state's serial number below)
  static_assert(kStepMaxPoints / kStepPoints <= kSolveThreads, "a column per lane");
  const uint32_t col = tid < ngroups ? tid : 0u;
  double tag_a = __hip_atomic_load(partA_in + (size_t)kAccN * ngroups + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  double tag_b = PL ? __hip_atomic_load(partB_in + (size_t)kGenN * ngroups + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
  MH_PHASE(0);
  // The state block this launch is meant to read carries the serial number `expect` -- (the alignment's epoch << 22) + the
  // launches before this one; a launch replayed from a captured graph is told its place in the chunk and adds the serial number
  // the host wrote into the parameter block before the replay -- written by the upload or by workgroup 0 of the previous launch.  Anything else in the block is older (the previous alignment's, the launch before
  // last's: the same buffer): wait for the right one rather than act on it.
  if (expect_rel) expect += __hip_atomic_load(&sk->step_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (uint32_t spins = 0;; spins++) {
    if (tid < kStateHeadDwords) lst_raw[tid] = __hip_atomic_load(reinterpret_cast<const uint32_t*>(s_in) + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (lst->serial == expect) break;
    if (spins == (1u << 14)) {  // ~ tens of milliseconds: give up loudly (the host fails the alignment)
      if (tid == 0) {
        atomicAdd(&s_canon->handover_timeouts, 1u);
        if (atomicCAS(&s_canon->dbg[0], 0u, 1u) == 0u) {
          s_canon->dbg[1] = wg; s_canon->dbg[2] = tid; s_canon->dbg[3] = expect; s_canon->dbg[4] = lst->serial; s_canon->dbg[5] = ngroups;
        }
      }
      break;
    }
    __builtin_amdgcn_s_sleep(8);
    __syncthreads();  // (lst_raw is rewritten)
  }
  // (taken from the block NOW: workgroup 0 rewrites both words further down, behind a barrier -- a wave that looked at
  //  `pending` after thread 0 had set it for the NEXT launch waited for partial sums nobody had written: launch 0 of an
  //  alignment "gave up" in workgroup 0, a few times per thousand alignments, more under load)
  const uint32_t serial = lst->serial;  // what the columns this launch sums must be tagged with
  const uint32_t pending = lst->pending;
  if (lst->done) {  // the loop has ended (the canonical block has it): keep the ping-pong consistent, nothing else
    if (wg == 0 && tid < kStateHeadDwords)
      __hip_atomic_store(reinterpret_cast<uint32_t*>(s_out) + tid, tid == kStateSerialDword ? serial + 1u : lst_raw[tid], __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  MH_PHASE(1);
  f32x4 stored = (f32x4){0.f, 0.f, 0.f, 0.f}, stored_c = stored, stored_n = stored;
  uint32_t stored_g = kNoMatch;
  bool have_stored = false;
  if (pending) {
    // every column carries the serial number of the launch that wrote it, stored AFTER its sums were acknowledged: a column
    // that does not carry this launch's number yet has not arrived (never seen since the exchange is at agent scope; a lane
    // waits for its columns rather than sum what is not there)
    const double want = (double)serial;
    for (uint32_t spins = 0; tag_a != want || (PL && tag_b != want); spins++) {
      if (spins == (1u << 16)) {
        atomicAdd(&s_canon->handover_timeouts, 1u);
        if (atomicCAS(&s_canon->dbg[0], 0u, 2u) == 0u) {
          s_canon->dbg[1] = wg; s_canon->dbg[2] = tid; s_canon->dbg[3] = serial; s_canon->dbg[4] = (uint32_t)tag_a; s_canon->dbg[5] = ngroups;
          s_canon->dbg[6] = expect;
          s_canon->dbg[7] = (pending & 0xFFu) | ((lst->iter & 0xFFu) << 8) | ((lst->inner & 0xFFu) << 16) | ((lst->done & 0xFFu) << 24);
        }
        break;
      }
      __builtin_amdgcn_s_sleep(1);
      tag_a = __hip_atomic_load(partA_in + (size_t)kAccN * ngroups + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (PL) tag_b = __hip_atomic_load(partB_in + (size_t)kGenN * ngroups + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    // (the sums' loads were also tried ahead of the state block, with the tags: 0.594 against 0.563 ms of ICP per scan -- ten more
    //  loads in front of the one the launch waits for)
    RowLoads<kAccN> ra;
    RowLoads<PL ? kGenN : 1> rb;
    rows_issue<kAccN>(ra, partA_in, ngroups, ngroups);
    if (PL) rows_issue<PL ? kGenN : 1>(rb, partB_in, ngroups, ngroups);
    // every column of the previous launch is tagged: what its workgroups stored for their groups is acknowledged (the tag is
    // written after that) -- the stored pairings of this lane's point are requested now, at agent scope
    // and BEHIND the loads of the sums this launch waits for, and arrive while the sums are formed and the step is solved
    stored = load_agent_b128(b_pair, ic);
    stored_g = __hip_atomic_load(pair_gidx + ic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (PL) {
      stored_c = load_agent_b128(b_plc, ic);
      stored_n = load_agent_b128(b_pln, ic);
    }
    have_stored = true;
    rows_finish(ra, ngroups, sh.totA, sh.red);
    if (PL) rows_finish(rb, ngroups, sh.totB, sh.red);
    solve_body<true, false>(lst, sk, nullptr, 0u, 0u, nullptr, 0u, 0u, sh, true, PL);
    __syncthreads();
  }
  const uint32_t done = lst->done;
  const bool body = !done && !close_only;
  if (wg == 0) {
    __syncthreads();  // every wave has taken `serial` and `pending` from the block
    if (tid == 0) {
      lst->pending = body ? 1u : 0u;
      lst->serial = serial + 1u;
    }
    __syncthreads();
    if (tid < kStateHeadDwords) {
      const uint32_t w = lst_raw[tid];
      __hip_atomic_store(reinterpret_cast<uint32_t*>(s_out) + tid, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (done && s_canon != s_out) G(reinterpret_cast<uint32_t*>(s_canon))[tid] = w;
    }
    if (tid == 0) {
      uint32_t* hp = sk->host_progress;
      if (hp) __hip_atomic_store(hp, (lst->iter & 0x7FFFFFFFu) | (done ? 0x80000000u : 0u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  MH_PHASE(11);
  if (!body) return;
  MH_PHASE(14);
  // What the previous launch produced is consumed with care.  Under load from other streams (sixteen sequences in one
  // process) a launch can begin up to ~0.25 us before the end time stamp of its predecessor on the same stream (rocprofv3
  // kernel trace: 10 of 3464 k_step16_b dispatches), and data the predecessor wrote last was seen stale by loads issued
  // first thing: a partial column two launches old -- ulp-sized differences between a batch and the same alignment alone,
  // a few per 200-scan run, gone with ANY extra microsecond before the reads.  Hence: state and partials cross launches
  // through agent-scope stores and loads (write-through; never answered from a stale L2 line), every partial column is
  // tagged with its launch's serial number once its sums are acknowledged and a reader waits for the tag it expects (the
  // tags are on their way before the state is known: no extra round trip), and the stored pairings -- the previous pairing bounds the search at an iteration start and IS the pairing at an
  // inner step -- are read here, microseconds into the launch, not prefetched at its top.
  if (!have_stored) {  // (no step was pending: what is stored is at least two launches old)
    stored = load_agent_b128(b_pair, ic);
    stored_g = __hip_atomic_load(pair_gidx + ic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (PL) {
      stored_c = load_agent_b128(b_plc, ic);
      stored_n = load_agent_b128(b_pln, ic);
    }
  }

  typedef const MatchK __attribute__((address_space(4))) * cmatchk_ptr;
  typedef const double __attribute__((address_space(4))) * cf64_ptr;
  const cmatchk_ptr ck = (cmatchk_ptr)uniform_const_ptr(kp);
  const uint32_t inner = lst->inner, iter = lst->iter;
  double T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = lst->T[k];
  const float thr2 = lst->cur_thr2, ang2 = lst->cur_ang2;
  const double kparam = lst->cur_kparam;
  const uint32_t kernel = ck->kernel;
  for (;;) {  // the groups of this workgroup (workgroup-uniform trip count)
    Acc a;
    acc_zero(a);
    double v[PL ? kGenN : 1];
#pragma unroll
    for (int j = 0; j < (PL ? kGenN : 1); j++) v[j] = 0.0;
    if (i < n) {  // row-uniform
      f32x4 q = stored, bc = (f32x4){0.f, 0.f, 0.f, 0.f}, bn = (f32x4){0.f, 0.f, 0.f, 0.f};
      bool ok, okp = false;
      if (inner == 0) {  // (workgroup-uniform) a new ICP iteration: the matchers
        float px, py, pz;
        transform_point(T, x, y, z, px, py, pz);
        float bound0 = __builtin_inff();
        if (iter > 0 && !map.no_prev_bound && stored.w < __builtin_inff()) {
          const float dx = stored.x - px, dy = stored.y - py, dz = stored.z - pz;
          bound0 = (dx * dx + dy * dy) + dz * dz;  // the candidate arithmetic of the scans
        }
        const NNResult r = nn_search_row16(map, r16, px, py, pz, bound0);
        const float n2 = (px * px + py * py) + pz * pz;
        ok = r.found && (r.d2 < thr2 + ang2 * n2);
        if (PL) {  // Matcher_Point2Plane first (k_match16_body)
          const float pl_thr = (float)((cf64_ptr)uniform_const_ptr(ck->pl_thr))[iter];
          okp = pl_row_search(map, r16, px, py, pz, pl_thr, bc, bn);
          if (r16 == 0) {
            store_agent_b128(b_plc, i, (f32x4){bc.x, bc.y, bc.z, okp ? 1.f : 0.f});
            store_agent_b128(b_pln, i, (f32x4){bn.x, bn.y, bn.z, 0.f});
          }
          if (okp && ck->skip_pl_paired) ok = false;  // (the nearest point still goes to pair_q: it bounds the next search)
        }
        if (r16 == 0) {
          store_agent_b128(b_pair, i, (f32x4){r.pt.x, r.pt.y, r.pt.z, r.d2});
          __hip_atomic_store(pair_gidx + i, ok ? __float_as_uint(r.pt.w) : kNoMatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        q = (f32x4){r.pt.x, r.pt.y, r.pt.z, r.d2};
      } else {  // an inner Gauss-Newton step: the stored pairings under the new pose
        ok = stored_g != kNoMatch;
        if (PL) {
          bc = stored_c;
          bn = stored_n;
          okp = bc.w != 0.f;
        }
      }
      if (r16 == 0) {
        acc_pt2pt_masked(a, T, ok, x, y, z, q.x, q.y, q.z, kernel, kparam, ck->w_pt2pt);
        if (PL && okp)
          acc_pt2pl_rows(v, T, x, y, z, make_float4(bc.x, bc.y, bc.z, 1.f), make_float4(bn.x, bn.y, bn.z, 0.f), kernel, kparam,
                         ck->w_pt2pl);
      }
    }
    MH_PHASE(12);
    // 32 row leaders -> one partial per sum and group, fixed order
    if (r16 == 0) {
#pragma unroll
      for (int j = 0; j < kAccN; j++) rowsA[j][row] = a.v[j];
      if (PL) {
#pragma unroll
        for (int j = 0; j < kGenN; j++) rowsB[PL ? j : 0][row] = v[PL ? j : 0];
      }
    }
    __syncthreads();
    if (tid < kAccN) {
      double sum = rowsA[tid][0];
#pragma unroll
      for (int r = 1; r < (int)kStepPoints; r++) sum += rowsA[tid][r];
      __hip_atomic_store(partA_out + tid * ngroups + g, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (PL && tid >= 64 && tid < 64 + kGenN) {  // (the second wave: the 29 point-to-plane sums)
      const uint32_t t = tid - 64;
      double sum = rowsB[PL ? t : 0][0];
#pragma unroll
      for (int r = 1; r < (int)kStepPoints; r++) sum += rowsB[PL ? t : 0][r];
      __hip_atomic_store(partB_out + t * ngroups + g, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // A column is tagged once its sums AND the pairings every wave of the workgroup stored for the group are acknowledged (a
    // reader that has seen the tag reads them): every wave counts itself in when its own stores are -- the pairings were stored
    // before the sums were formed, so this adds nothing to what the tagging wave waits for anyway -- and the tagging wave waits
    // for the count.
    acks_wanted += kSolveThreads / 64u;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if ((tid & 63u) == 0u) __hip_atomic_fetch_add(&pair_acks, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (tid < 64 || (PL && tid < 128)) {  // (the first wave holds the 18 sums: lane 18 tags the column; the second wave's lane 29 the other kind's)
      while (__hip_atomic_load(&pair_acks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < acks_wanted) __builtin_amdgcn_s_sleep(1);
      if (tid == kAccN) __hip_atomic_store(partA_out + (size_t)kAccN * ngroups + g, (double)(serial + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (PL && tid == 64 + kGenN) __hip_atomic_store(partB_out + (size_t)kGenN * ngroups + g, (double)(serial + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    g += nw;
    if (g >= ngroups) break;
    i = g * kStepPoints + row;
    ic = i < n ? i : n - 1;
    x = G(lx)[ic]; y = G(ly)[ic]; z = G(lz)[ic];
    stored = load_agent_b128(b_pair, ic);
    stored_g = __hip_atomic_load(pair_gidx + ic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (PL) {
      stored_c = load_agent_b128(b_plc, ic);
      stored_n = load_agent_b128(b_pln, ic);
    }
    __syncthreads();  // (the row buffers are reused)
  }
  MH_PHASE(13);
}
template <bool PL>
__global__ __launch_bounds__(kSolveThreads) void k_step16(const IcpDeviceState* __restrict__ s_in, IcpDeviceState* s_out,
                                                          IcpDeviceState* s_canon, const MatchK* __restrict__ kp,
                                                          const SolveK* __restrict__ sk, const float* __restrict__ lx,
                                                          const float* __restrict__ ly, const float* __restrict__ lz,
                                                          uint32_t n, MapView map, float4* pair_q, uint32_t* pair_gidx,
                                                          float4* pl_c, float4* pl_n, const double* __restrict__ partA_in,
                                                          double* __restrict__ partA_out, const double* __restrict__ partB_in,
                                                          double* __restrict__ partB_out, uint32_t ngroups, uint32_t close_only,
                                                          uint32_t expect, uint32_t expect_rel) {
  k_step16_body<PL>(s_in, s_out, s_canon, kp, sk, lx, ly, lz, n, map, pair_q, pair_gidx, pl_c, pl_n, partA_in, partA_out, partB_in,
                    partB_out, ngroups, gridDim.x, close_only, expect, expect_rel);
}
// in lock step: blockIdx.y = job; `par`: which state block / partials half this launch reads; gridDim.x: the host's cap on a
// job's workgroups
template <bool PL>
__global__ __launch_bounds__(kSolveThreads) void k_step16_b(const BatchJob* __restrict__ jobs, uint32_t src, uint32_t par, uint32_t close_only,
                                                            uint32_t launch_index) {
  const BatchJob& j = jobs[blockIdx.y];
  const uint32_t ng0 = (j.n + kStepPoints - 1) / kStepPoints;
  const uint32_t ngroups = ng0 ? ng0 : 1u;
  IcpDeviceState* const S[3] = {j.st_b, reinterpret_cast<IcpDeviceState*>(reinterpret_cast<char*>(j.st_b) + 256), j.st};
  const uint32_t dst = close_only ? 2u : (src == 2u ? 0u : (src ^ 1u));
  double* const pa[2] = {j.part, j.part + (size_t)kStepRowsA * ngroups};
  double* const pb[2] = {j.partb, j.partb ? j.partb + (size_t)kStepRowsB * ngroups : nullptr};
  k_step16_body<PL>(S[src], S[dst], S[2], j.mk, j.sk, j.lx, j.ly, j.lz, j.n, j.map, j.pair_q, j.pair_gidx,
                    j.pl_c, j.pl_n, pa[par], pa[par ^ 1u], pb[par], pb[par ^ 1u], ngroups, ngroups < gridDim.x ? ngroups : gridDim.x,
                    close_only, j.serial_base + launch_index, 0u);
}

// ================================================================================================
// k_icp16: the small layer's WHOLE loop in one launch (round 5) -- the k_step16 chain without its launch boundaries.
// One workgroup per group of 32 points, all of them resident for the duration (the host admits a loop only while the
// workgroups of all running loops fit the part's CUs, and falls back to the chain otherwise or when a workgroup gives up
// waiting); every workgroup keeps its own copy of the state block in LDS and closes every Gauss-Newton step itself -- the
// same ordered sums and the same solve_body as k_step16, the same bits -- so that only the partial sums cross workgroups:
//   body (search or re-accumulate, the pairings of the group stay in registers) -> the group's column of sums, every sum a
//   16-byte entry {value, serial number, check word} in ONE agent-scope store -> every workgroup loads all columns of the
//   step and retries the entries that do not carry the step's serial number yet (no separate tag: one round trip instead of
//   store | acknowledge | tag | poll | load) -> ordered sums -> solve -> next body.
// tools/xcd_exchange.hip prices the exchange alone: 2.1 us for 44 workgroups x 18 sums (3.2 with the 29 plane sums; 3.1 / 4+
// with a tag per column), against ~4.5 us of launch boundary + ~2 us of tagged exchange per k_step16 launch.  (The same tool:
// workgroups of a launch are dealt to the XCDs round-robin, blockIdx % 8, but sc0 loads do not bypass the L1 -- an exchange
// confined to one XCD's L2 has no cheaper load than the agent-scope one, and 32 CUs would hold 32 groups only.)
// The entries ping-pong between two halves by step parity: a workgroup writes step s + 2's entries after it has summed
// step s + 1, which every workgroup wrote after reading step s.  Serial numbers never repeat within a context (a counter
// advanced by every loop's step budget), so an entry of an earlier alignment is never taken for the current one.
// ================================================================================================
#ifndef MH_LOOP_MAX_GROUPS
#define MH_LOOP_MAX_GROUPS 80  // 2560 points: every ICP layer of the city drive (<= 2311 points; 15 % of them above 2048) -- and the last size at which a
                               // lane of the 18-row fetch still loads three entries (28 columns per lane group); 64 until round 6: +6 % on the single sequence
#endif
constexpr uint32_t kLoopMaxGroups = MH_LOOP_MAX_GROUPS;   // workgroups of one k_icp16 loop: layers up to 32 x this many points (beyond: the chain)
constexpr uint32_t kLwMaxGroups = 128;                    // columns of one k_icpw loop (mh_loop_wave.h): layers up to 4096 points
constexpr uint32_t kLoopRowStride = kLwMaxGroups > kLoopMaxGroups ? kLwMaxGroups : kLoopMaxGroups;   // entries from one sum's row to the next
constexpr size_t kLoopExchangeBytes = 2 * (size_t)(kAccN + kGenN) * kLoopRowStride * 16;

__device__ __forceinline__ void cov_prepare_lane(const Pose& Tc, int j, double hx, double ha, double* out);  // (below)

constexpr unsigned long long kLoopDeadlineTicks = 2000000ull;  // 20 ms of the 100 MHz wall clock: a loop's workgroups that do not meet by then give up
template <int NVALS>
__device__ __forceinline__ void loop_rows_fetch(RowLoads<NVALS, (int)kLoopMaxGroups>& r, const AgentBuf& x, uint32_t base, uint32_t n,
                                                uint32_t serial, uint32_t* gave_up, unsigned long long deadline) {
  typedef RowLoads<NVALS, (int)kLoopMaxGroups> RL;
  const int row = (int)threadIdx.x / RL::kG, g = (int)threadIdx.x % RL::kG;
  uint32_t need = 0;
#pragma unroll
  for (int j = 0; j < RL::kL; j++) {
    r.v[j] = 0.0;
    if (row < NVALS && (uint32_t)(g + j * RL::kG) < n) need |= 1u << j;
  }
  const uint32_t e0 = base + (uint32_t)(row < NVALS ? row : 0) * kLoopRowStride + (uint32_t)g;
  for (uint32_t spins = 0; need; spins++) {
    u32x4v w[RL::kL];
#pragma unroll
    for (int j = 0; j < RL::kL; j++)
      if ((need >> j) & 1u) w[j] = __builtin_amdgcn_raw_buffer_load_b128(x.rsrc, (int)((e0 + (uint32_t)(j * RL::kG)) * 16u), 0, /*aux: sc1*/ 16);
#pragma unroll
    for (int j = 0; j < RL::kL; j++)
      if (((need >> j) & 1u) && w[j].z == serial && w[j].w == (w[j].x ^ w[j].y ^ w[j].z)) {
        r.v[j] = __hiloint2double((int)w[j].y, (int)w[j].x);
        need &= ~(1u << j);
      }
    if (!need) break;
    // (a wall-clock limit, not a spin count -- ADVICE r5: what a spin costs depends on who else is on the device)
    if (((spins & 31u) == 31u && wall_clock64() > deadline) || __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) {  // 20 ms: give up loudly
      __hip_atomic_store(gave_up, 1u + (uint32_t)row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}
__device__ __forceinline__ void loop_entry_store(const AgentBuf& x, uint32_t e, double v, uint32_t serial) {
  const uint32_t lo = (uint32_t)__double2loint(v), hi = (uint32_t)__double2hiint(v);
  const u32x4v w = {lo, hi, serial, lo ^ hi ^ serial};
  __builtin_amdgcn_raw_buffer_store_b128(w, x.rsrc, (int)(e * 16u), 0, /*aux: sc1*/ 16);
}

template <bool PL>
__global__ __launch_bounds__(kSolveThreads) void k_icp16(IcpDeviceState* s_canon, const MatchK* __restrict__ kp,
                                                         const SolveK* __restrict__ sk, const float* __restrict__ lx,
                                                         const float* __restrict__ ly, const float* __restrict__ lz, uint32_t n,
                                                         MapView map, float4* pair_q, uint32_t* pair_gidx, float4* pl_c, float4* pl_n,
                                                         void* xa, void* xb, uint32_t ngroups, uint32_t serial0, uint32_t max_steps,
                                                         uint32_t want_cov) {
  __shared__ SolveShared sh;
  __shared__ __attribute__((aligned(8))) uint32_t lst_raw[kStateHeadDwords];
  __shared__ double rowsA[kAccN][kStepPoints + 1];
  __shared__ double rowsB[PL ? kGenN : 1][kStepPoints + 1];
  __shared__ uint32_t gave_up;
  const uint32_t tid = threadIdx.x, g = blockIdx.x;
  if (g >= ngroups) return;
  IcpDeviceState* const lst = reinterpret_cast<IcpDeviceState*>(lst_raw);
  const AgentBuf bxa = agent_buf(xa, 2u * kAccN * kLoopRowStride), bxb = agent_buf(PL ? xb : xa, 2u * (PL ? kGenN : kAccN) * kLoopRowStride);
  const uint32_t row = tid >> 4, r16 = tid & 15u;
  const uint32_t i = g * kStepPoints + row, ic = i < n ? i : n - 1;
  const float x = G(lx)[ic], y = G(ly)[ic], z = G(lz)[ic];
  if (tid < kStateHeadDwords) lst_raw[tid] = G(reinterpret_cast<const uint32_t*>(s_canon))[tid];  // (uploaded before the launch)
  if (tid == 0) gave_up = 0;
  __syncthreads();
  typedef const MatchK __attribute__((address_space(4))) * cmatchk_ptr;
  typedef const double __attribute__((address_space(4))) * cf64_ptr;
  const cmatchk_ptr ck = (cmatchk_ptr)uniform_const_ptr(kp);
  const uint32_t kernel = ck->kernel;
  const unsigned long long deadline = wall_clock64() + kLoopDeadlineTicks;
  // the pairing of this row's point: found at an iteration's start, used by its inner steps and as the next search's bound
  f32x4 q = (f32x4){0.f, 0.f, 0.f, __builtin_inff()}, bc = (f32x4){0.f, 0.f, 0.f, 0.f}, bn = (f32x4){0.f, 0.f, 0.f, 0.f};
  bool ok = false, okp = false;
  uint32_t step = 0;
  MH_LOOP_STAMPS;
#pragma nounroll
  for (;; step++) {
    MH_LOOP_STAMP(0);
    if (lst->pending) {  // the sums of step - 1 (serial0 + step), every workgroup for itself
      const uint32_t half = (step - 1u) & 1u;
      RowLoads<kAccN, (int)kLoopMaxGroups> ra;
      RowLoads<PL ? kGenN : 1, (int)kLoopMaxGroups> rb;
      loop_rows_fetch<kAccN>(ra, bxa, half * kAccN * kLoopRowStride, ngroups, serial0 + step, &gave_up, deadline);
      if (PL) loop_rows_fetch<PL ? kGenN : 1>(rb, bxb, half * kGenN * kLoopRowStride, ngroups, serial0 + step, &gave_up, deadline);
      MH_LOOP_STAMP(1);
      rows_finish(ra, ngroups, sh.totA, sh.red);
      if (PL) rows_finish(rb, ngroups, sh.totB, sh.red);
      MH_LOOP_STAMP(2);
      if (gave_up) break;  // (behind rows_finish' barriers: the same in every wave)
      solve_body<true, false>(lst, sk, nullptr, 0u, 0u, nullptr, 0u, 0u, sh, true, PL);
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
    double v[PL ? kGenN : 1];
#pragma unroll
    for (int j = 0; j < (PL ? kGenN : 1); j++) v[j] = 0.0;
    if (i < n) {  // row-uniform
      if (inner == 0) {  // (workgroup-uniform) a new ICP iteration: the matchers, exactly k_step16's
        float px, py, pz;
        transform_point(T, x, y, z, px, py, pz);
        float bound0 = __builtin_inff();
        if (iter > 0 && !map.no_prev_bound && q.w < __builtin_inff()) {
          const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
          bound0 = (dx * dx + dy * dy) + dz * dz;  // the candidate arithmetic of the scans
        }
        MH_LOOP_STAMP(6);
        const NNResult r = nn_search_row16(map, r16, px, py, pz, bound0);
        MH_LOOP_STAMP(7);
        const float n2 = (px * px + py * py) + pz * pz;
        ok = r.found && (r.d2 < thr2 + ang2 * n2);
        if (PL) {  // Matcher_Point2Plane first (k_match16_body)
          const float pl_thr = (float)((cf64_ptr)uniform_const_ptr(ck->pl_thr))[iter];
          okp = pl_row_search(map, r16, px, py, pz, pl_thr, bc, bn);
          if (r16 == 0) {  // (read by the covariance kernels and the pairing export once the loop has ended)
            pl_c[i] = make_float4(bc.x, bc.y, bc.z, okp ? 1.f : 0.f);
            pl_n[i] = make_float4(bn.x, bn.y, bn.z, 0.f);
          }
          if (okp && ck->skip_pl_paired) ok = false;  // (the nearest point still bounds the next search)
        }
        if (r16 == 0) {
          pair_q[i] = make_float4(r.pt.x, r.pt.y, r.pt.z, r.d2);
          G(pair_gidx)[i] = ok ? __float_as_uint(r.pt.w) : kNoMatch;
        }
        q = (f32x4){r.pt.x, r.pt.y, r.pt.z, r.d2};
      }
      if (r16 == 0) {
        acc_pt2pt_masked(a, T, ok, x, y, z, q.x, q.y, q.z, kernel, kparam, ck->w_pt2pt);
        if (PL && okp)
          acc_pt2pl_rows(v, T, x, y, z, make_float4(bc.x, bc.y, bc.z, 1.f), make_float4(bn.x, bn.y, bn.z, 0.f), kernel, kparam,
                         ck->w_pt2pl);
      }
    }
    if (r16 == 0) {
#pragma unroll
      for (int j = 0; j < kAccN; j++) rowsA[j][row] = a.v[j];
      if (PL) {
#pragma unroll
        for (int j = 0; j < kGenN; j++) rowsB[PL ? j : 0][row] = v[PL ? j : 0];
      }
    }
    MH_LOOP_STAMP(8);
    __syncthreads();
    MH_LOOP_STAMP(4);
    const uint32_t out = step & 1u;
    if (tid < kAccN) {
      double sum = rowsA[tid][0];
#pragma unroll
      for (int r = 1; r < (int)kStepPoints; r++) sum += rowsA[tid][r];
      loop_entry_store(bxa, (out * kAccN + tid) * kLoopRowStride + g, sum, serial0 + step + 1u);
    }
    if (PL && tid >= 64 && tid < 64 + kGenN) {  // (the second wave: the 29 point-to-plane sums)
      const uint32_t t = tid - 64;
      double sum = rowsB[PL ? t : 0][0];
#pragma unroll
      for (int r = 1; r < (int)kStepPoints; r++) sum += rowsB[PL ? t : 0][r];
      loop_entry_store(bxb, (out * kGenN + t) * kLoopRowStride + g, sum, serial0 + step + 1u);
    }
    if (tid == 0) lst->pending = 1u;
    __syncthreads();
    MH_LOOP_STAMP(5);
  }
  MH_LOOP_STAMPS_OUT(step);
  if (gave_up) {  // the canonical block keeps done == 0: the host runs the alignment again, launch by launch
    if (tid == 0) {
      atomicAdd(&s_canon->handover_timeouts, 1u);
      if (atomicCAS(&s_canon->dbg[0], 0u, 3u) == 0u) {
        s_canon->dbg[1] = g; s_canon->dbg[2] = gave_up - 1u; s_canon->dbg[3] = serial0 + step; s_canon->dbg[4] = step; s_canon->dbg[5] = ngroups;
      }
    }
    return;
  }
  if (g == 0) {
    if (tid < kStateHeadDwords) G(reinterpret_cast<uint32_t*>(s_canon))[tid] = lst_raw[tid];
    if (want_cov && lst->done && tid < 6) {  // k_cov_prepare's six lanes: the covariance chain that follows starts at k_cov_accum
      Pose Tc;
#pragma unroll
      for (int k = 0; k < 12; k++) Tc.m[k] = lst->T[k];
      double out[12];
      cov_prepare_lane(Tc, (int)tid, sk->cov_hx, sk->cov_ha, out);
#pragma unroll
      for (int k = 0; k < 12; k++) s_canon->covD[tid * 12 + k] = out[k];
    }
  }
}

// k_icp16_b: the same loop for the jobs of a lock-step group, side by side in ONE launch (blockIdx.y = job) -- every job's
// workgroups exchange among themselves only; a workgroup takes the groups x, x + nw, ... of its job (the host caps nw so that the
// workgroups of ALL jobs are resident together), one column of sums per GROUP as everywhere: the same bits.  The pairings of a
// workgroup's groups wait in LDS (up to kLoopGroupsPerWg groups of 32 rows) instead of registers.
constexpr uint32_t kLoopGroupsPerWg = 8;
template <bool PL>
__device__ __forceinline__ void icp16_multi_body(const BatchJob& j) {
  __shared__ SolveShared sh;
  __shared__ __attribute__((aligned(8))) uint32_t lst_raw[kStateHeadDwords];
  __shared__ double rowsA[kAccN][kStepPoints + 1];
  __shared__ double rowsB[PL ? kGenN : 1][kStepPoints + 1];
  __shared__ f32x4 keep_q[kLoopGroupsPerWg][kStepPoints];
  __shared__ f32x4 keep_c[PL ? kLoopGroupsPerWg : 1][kStepPoints], keep_n[PL ? kLoopGroupsPerWg : 1][kStepPoints];
  __shared__ uint32_t keep_ok[kLoopGroupsPerWg][kStepPoints];  // bit 0: point pairing accepted, bit 1: plane pairing
  __shared__ uint32_t gave_up;
  const uint32_t n = j.n;
  const uint32_t ng0 = (n + kStepPoints - 1) / kStepPoints, ngroups = ng0 ? ng0 : 1u;
  const uint32_t nw = ngroups < gridDim.x ? ngroups : gridDim.x;
  const uint32_t tid = threadIdx.x, wg = blockIdx.x;
  if (wg >= nw || n == 0) return;
  IcpDeviceState* const s_canon = j.st;
  const SolveK* const sk = j.sk;
  const MapView map = j.map;
  IcpDeviceState* const lst = reinterpret_cast<IcpDeviceState*>(lst_raw);
  const AgentBuf bxa = agent_buf(j.loop_xa, 2u * kAccN * kLoopRowStride), bxb = agent_buf(PL ? j.loop_xb : j.loop_xa, 2u * (PL ? kGenN : kAccN) * kLoopRowStride);
  const uint32_t row = tid >> 4, r16 = tid & 15u;
  const uint32_t serial0 = j.loop_serial0;
  if (tid < kStateHeadDwords) lst_raw[tid] = G(reinterpret_cast<const uint32_t*>(s_canon))[tid];  // (scattered before the launch)
  if (tid == 0) gave_up = 0;
  __syncthreads();
  if (lst->done) return;  // (a job that was finished before the batch began: nothing to do)
  typedef const MatchK __attribute__((address_space(4))) * cmatchk_ptr;
  typedef const double __attribute__((address_space(4))) * cf64_ptr;
  const cmatchk_ptr ck = (cmatchk_ptr)uniform_const_ptr(j.mk);
  const uint32_t kernel = ck->kernel;
  const unsigned long long deadline = wall_clock64() + kLoopDeadlineTicks;
  const uint32_t max_steps = j.loop_pad ? 1u : sk->max_iterations * sk->max_inner + 1u;  // (loop_pad: MH_LOOP16_TEST_ABANDON, the loop is cut short)
  uint32_t step = 0;
#pragma nounroll
  for (;; step++) {
    if (lst->pending) {
      const uint32_t half = (step - 1u) & 1u;
      RowLoads<kAccN, (int)kLoopMaxGroups> ra;
      RowLoads<PL ? kGenN : 1, (int)kLoopMaxGroups> rb;
      loop_rows_fetch<kAccN>(ra, bxa, half * kAccN * kLoopRowStride, ngroups, serial0 + step, &gave_up, deadline);
      if (PL) loop_rows_fetch<PL ? kGenN : 1>(rb, bxb, half * kGenN * kLoopRowStride, ngroups, serial0 + step, &gave_up, deadline);
      rows_finish(ra, ngroups, sh.totA, sh.red);
      if (PL) rows_finish(rb, ngroups, sh.totB, sh.red);
      if (gave_up) break;
      solve_body<true, false>(lst, sk, nullptr, 0u, 0u, nullptr, 0u, 0u, sh, true, PL);
      if (tid == 0) lst->pending = 0u;
      __syncthreads();
    }
    if (lst->done || step >= max_steps) break;
    const uint32_t inner = lst->inner, iter = lst->iter;
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = lst->T[k];
    const float thr2 = lst->cur_thr2, ang2 = lst->cur_ang2;
    const double kparam = lst->cur_kparam;
    const uint32_t out = step & 1u;
    uint32_t slot = 0;
#pragma nounroll
    for (uint32_t g = wg; g < ngroups; g += nw, slot++) {  // (workgroup-uniform trip count)
      const uint32_t i = g * kStepPoints + row, ic = i < n ? i : n - 1;
      const float x = G(j.lx)[ic], y = G(j.ly)[ic], z = G(j.lz)[ic];
      Acc a;
      acc_zero(a);
      double v[PL ? kGenN : 1];
#pragma unroll
      for (int q = 0; q < (PL ? kGenN : 1); q++) v[q] = 0.0;
      if (i < n) {
        f32x4 q = keep_q[slot][row], bc = (f32x4){0.f, 0.f, 0.f, 0.f}, bn = (f32x4){0.f, 0.f, 0.f, 0.f};
        bool ok, okp = false;
        if (inner == 0) {
          float px, py, pz;
          transform_point(T, x, y, z, px, py, pz);
          float bound0 = __builtin_inff();
          if (iter > 0 && !map.no_prev_bound && q.w < __builtin_inff()) {
            const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
            bound0 = (dx * dx + dy * dy) + dz * dz;
          }
          const NNResult r = nn_search_row16(map, r16, px, py, pz, bound0);
          const float n2 = (px * px + py * py) + pz * pz;
          ok = r.found && (r.d2 < thr2 + ang2 * n2);
          if (PL) {
            const float pl_thr = (float)((cf64_ptr)uniform_const_ptr(ck->pl_thr))[iter];
            okp = pl_row_search(map, r16, px, py, pz, pl_thr, bc, bn);
            if (r16 == 0) {
              j.pl_c[i] = make_float4(bc.x, bc.y, bc.z, okp ? 1.f : 0.f);
              j.pl_n[i] = make_float4(bn.x, bn.y, bn.z, 0.f);
              keep_c[PL ? slot : 0][row] = bc;
              keep_n[PL ? slot : 0][row] = bn;
            }
            if (okp && ck->skip_pl_paired) ok = false;
          }
          q = (f32x4){r.pt.x, r.pt.y, r.pt.z, r.d2};
          if (r16 == 0) {
            j.pair_q[i] = make_float4(r.pt.x, r.pt.y, r.pt.z, r.d2);
            G(j.pair_gidx)[i] = ok ? __float_as_uint(r.pt.w) : kNoMatch;
            keep_q[slot][row] = q;
            keep_ok[slot][row] = (ok ? 1u : 0u) | (okp ? 2u : 0u);
          }
        } else {
          const uint32_t f = keep_ok[slot][row];
          ok = (f & 1u) != 0u;
          okp = (f & 2u) != 0u;
          if (PL) {
            bc = keep_c[PL ? slot : 0][row];
            bn = keep_n[PL ? slot : 0][row];
          }
        }
        if (r16 == 0) {
          acc_pt2pt_masked(a, T, ok, x, y, z, q.x, q.y, q.z, kernel, kparam, ck->w_pt2pt);
          if (PL && okp)
            acc_pt2pl_rows(v, T, x, y, z, make_float4(bc.x, bc.y, bc.z, 1.f), make_float4(bn.x, bn.y, bn.z, 0.f), kernel, kparam,
                           ck->w_pt2pl);
        }
      }
      if (r16 == 0) {
#pragma unroll
        for (int q = 0; q < kAccN; q++) rowsA[q][row] = a.v[q];
        if (PL) {
#pragma unroll
          for (int q = 0; q < kGenN; q++) rowsB[PL ? q : 0][row] = v[PL ? q : 0];
        }
      }
      __syncthreads();
      if (tid < kAccN) {
        double sum = rowsA[tid][0];
#pragma unroll
        for (int r = 1; r < (int)kStepPoints; r++) sum += rowsA[tid][r];
        loop_entry_store(bxa, (out * kAccN + tid) * kLoopRowStride + g, sum, serial0 + step + 1u);
      }
      if (PL && tid >= 64 && tid < 64 + kGenN) {
        const uint32_t t = tid - 64;
        double sum = rowsB[PL ? t : 0][0];
#pragma unroll
        for (int r = 1; r < (int)kStepPoints; r++) sum += rowsB[PL ? t : 0][r];
        loop_entry_store(bxb, (out * kGenN + t) * kLoopRowStride + g, sum, serial0 + step + 1u);
      }
      __syncthreads();  // (the row buffers are reused by the workgroup's next group)
    }
    if (tid == 0) lst->pending = 1u;
    __syncthreads();
  }
  if (gave_up) {
    if (tid == 0) {
      atomicAdd(&s_canon->handover_timeouts, 1u);
      if (atomicCAS(&s_canon->dbg[0], 0u, 4u) == 0u) {
        s_canon->dbg[1] = wg; s_canon->dbg[2] = gave_up - 1u; s_canon->dbg[3] = serial0 + step; s_canon->dbg[4] = step; s_canon->dbg[5] = ngroups;
      }
    }
    return;
  }
  if (wg == 0 && tid < kStateHeadDwords) G(reinterpret_cast<uint32_t*>(s_canon))[tid] = lst_raw[tid];
}
template <bool PL>
__global__ __launch_bounds__(kSolveThreads) void k_icp16_b(const BatchJob* __restrict__ jobs) {
  icp16_multi_body<PL>(jobs[blockIdx.y]);
}

#include "mh_loop_wave.h"  // k_icpw: the same loop with the plan / scan search, 128 points per workgroup of 256 lanes
