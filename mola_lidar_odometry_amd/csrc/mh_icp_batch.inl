// mh_icp_batch.inl -- mh_icp_align_batch: many alignments (one per context) from one host thread, jobs of the same kernel chain
// advancing in lock step (one launch per kernel over all jobs), the whole loops of small layers side by side in one launch.
// Included by mh_icp.hip inside its extern "C" block (it launches the *_b kernels of that translation unit).

namespace {
// descriptor of one job for the *_b kernels (device pointers only; the pairing-block fields are filled by the caller)
void fill_batch_desc(const AlignJob& j, BatchJob& d) {
  memset(&d, 0, sizeof(d));
  d.st = j.ctx->d_state;
  d.st_b = j.ctx->d_state_b;
  d.serial_base = j.serial_base;
  d.mk = &j.ctx->d_params->mk;
  d.sk = &j.ctx->d_params->sk;
  d.lx = j.scan->x; d.ly = j.scan->y; d.lz = j.scan->z;
  d.n = (uint32_t)j.scan->n;
  d.nb = j.nb;
  d.nba = j.nba;
  d.nbm = j.nbm;
  d.map = j.map->view();
  d.pair_q = j.ctx->pair_q.as<float4>();
  d.pair_gidx = j.ctx->pair_gidx.as<uint32_t>();
  d.part = j.ctx->partials.as<double>();
  if (j.pl) {
    d.partb = j.ctx->partials_b.as<double>();
    d.pl_c = j.ctx->pl_c.as<float4>();
    d.pl_n = j.ctx->pl_n.as<float4>();
  }
  d.sched_dst = j.ctx->sched.as<uint32_t>();
  d.sched_dwords = (uint32_t)(2 * j.nsched_pending);
  if (j.variant >= 6 && j.variant != 9) {
    d.sx = j.scan->sx; d.sy = j.scan->sy; d.sz = j.scan->sz;
    d.perm = j.scan->perm;
    d.tile_start = j.scan->tile_start;
    d.n_tiles = j.scan->n_tiles;
  }
}

// Work still queued on a job's own stream (asynchronous uploads, filters, de-skew, an earlier alignment) must be
// complete before `lead`'s stream reads that job's scan / state: an event per job, waited for by the leader's stream.
mh_status order_after_job_streams(mh_ctx* lead, const std::vector<AlignJob*>& jobs) {
  for (AlignJob* j : jobs) {
    MH_TRY(map_ready_on(j->map, lead->stream));  // a key-frame update of this job's map still running on its side stream
    if (j->ctx == lead || j->ctx->stream == lead->stream) continue;
    if (hipStreamQuery(j->ctx->stream) == hipSuccess) continue;  // nothing pending there
    MH_HIP(hipEventRecord(j->ctx->ev_ready, j->ctx->stream));
    MH_HIP(hipStreamWaitEvent(lead->stream, j->ctx->ev_ready, 0));
  }
  (void)hipGetLastError();  // hipStreamQuery's hipErrorNotReady is not an error
  return MH_OK;
}

// Where the batch's final pairings go: device-side layout of the pairs block (header of per-block counts / offsets for
// the compaction + the block itself unless the caller's block already is device memory).
struct PairsPlan {
  bool want = false;
  int32_t mem = MH_MEM_HOST;
  char* host_block = nullptr;    // caller's block (host kinds)
  char* dev_block = nullptr;     // where the kernels write
  uint32_t* dev_hdr = nullptr;
  size_t total_bytes = 0;
  std::vector<size_t> off;       // byte offset of job i in the block
  std::vector<size_t> hdr_off;   // entry offset of job i's counts in the header
};

mh_status plan_pairs(mh_ctx* lead, const std::vector<AlignJob>& jobs, void* pairs_block, int32_t pairs_mem, PairsPlan& pp) {
  pp.want = pairs_block != nullptr;
  if (!pp.want) return MH_OK;
  pp.mem = pairs_mem;
  pp.off.resize(jobs.size());
  pp.hdr_off.resize(jobs.size());
  size_t bytes = 0, hdr = 0;
  for (size_t i = 0; i < jobs.size(); i++) {
    pp.off[i] = bytes;
    pp.hdr_off[i] = hdr;
    bytes += mh_pairs_block_bytes(jobs[i].scan->n);
    hdr += 2 * (size_t)nblk(jobs[i].scan->n ? jobs[i].scan->n : 1);
  }
  pp.total_bytes = bytes;
  const size_t hdr_bytes = (hdr * 4 + 255) / 256 * 256;
  const size_t need = hdr_bytes + (pairs_mem == MH_MEM_DEVICE ? 0 : bytes);
  if (lead->pairs_stage.bytes < need && lead->pairs_copy_pending) {  // the previous download still reads the old buffer
    MH_HIP(mh::wait_stream(lead->copy_stream));
    lead->pairs_copy_pending = false;
  }
  MH_TRY(lead->pairs_stage.reserve(need));
  pp.dev_hdr = lead->pairs_stage.as<uint32_t>();
  pp.dev_block = pairs_mem == MH_MEM_DEVICE ? (char*)pairs_block : lead->pairs_stage.as<char>() + hdr_bytes;
  pp.host_block = pairs_mem == MH_MEM_DEVICE ? nullptr : (char*)pairs_block;
  if (pairs_mem == MH_MEM_HOST_PINNED && !lead->copy_stream) {
    MH_HIP(hipStreamCreateWithFlags(&lead->copy_stream, hipStreamNonBlocking));
    MH_HIP(hipEventCreateWithFlags(&lead->ev_pairs_ready, hipEventDisableTiming));
    MH_HIP(hipEventCreateWithFlags(&lead->ev_pairs_copied, hipEventDisableTiming));
  }
  return MH_OK;
}

void set_pairs_fields(const PairsPlan& pp, size_t job_index, BatchJob& d) {
  if (!pp.want) return;
  d.cp_counts = pp.dev_hdr + pp.hdr_off[job_index];
  d.cp_out = reinterpret_cast<uint32_t*>(pp.dev_block + pp.off[job_index]);
  d.cp_stride = (uint32_t)(mh_pairs_block_bytes(d.n) / 24);
}

// compaction of every finished job's pairings into the block (descriptors `dj` already on the device) + the download
// `lead` owns the staging buffer, the copy stream and its events (plan_pairs: always the FIRST job's context, which is
// what mh_ctx_synchronize(scans[0]'s context) and the next batch wait on); `s` is the stream the compaction runs on --
// the lock-step group's, which is another context's when job 0 is trivial (same device: events order them).
mh_status finish_pairs(mh_ctx* lead, hipStream_t s, const PairsPlan& pp, const BatchJob* dj, uint32_t A, uint32_t gx_cov) {
  if (!pp.want || A == 0) return MH_OK;
  if (lead->pairs_copy_pending) MH_HIP(hipStreamWaitEvent(s, lead->ev_pairs_copied, 0));  // staging still being read
  hipLaunchKernelGGL(k_count_valid_b, dim3(gx_cov, A), dim3(kBlock), 0, s, dj);
  hipLaunchKernelGGL(k_scan_blocks_b, dim3(1, A), dim3(1024), 0, s, dj);
  hipLaunchKernelGGL(k_compact_b, dim3(gx_cov, A), dim3(kBlock), 0, s, dj);
  MH_HIP(hipGetLastError());
  if (pp.mem == MH_MEM_HOST) {
    MH_HIP(hipMemcpyAsync(pp.host_block, pp.dev_block, pp.total_bytes, hipMemcpyDeviceToHost, s));
    MH_HIP(mh::wait_stream(s));
  } else if (pp.mem == MH_MEM_HOST_PINNED) {
    // on the copy stream: the call returns, the next batch's kernels run while this block travels
    MH_HIP(hipEventRecord(lead->ev_pairs_ready, s));
    MH_HIP(hipStreamWaitEvent(lead->copy_stream, lead->ev_pairs_ready, 0));
    MH_HIP(hipMemcpyAsync(pp.host_block, pp.dev_block, pp.total_bytes, hipMemcpyDeviceToHost, lead->copy_stream));
    MH_HIP(hipEventRecord(lead->ev_pairs_copied, lead->copy_stream));
    lead->pairs_copy_pending = true;
  }
  return MH_OK;
}
}  // namespace

static mh_status align_batch_run(size_t n_jobs, const mh_map* const* maps, const mh_scan* const* scans,
                                 const mh_icp_params* params, int32_t params_per_job, const double* T_guesses,
                                 const mh_prior* const* priors, mh_icp_result* results, void* pairs_block, int32_t pairs_mem);

mh_status mh_icp_align_batch(size_t n_jobs, const mh_map* const* maps, const mh_scan* const* scans,
                             const mh_icp_params* params, int32_t params_per_job, const double* T_guesses,
                             const mh_prior* const* priors, mh_icp_result* results, void* pairs_block, int32_t pairs_mem) {
  const mh_status st = align_batch_run(n_jobs, maps, scans, params, params_per_job, T_guesses, priors, results, pairs_block, pairs_mem);
  // MH_DEBUG_VERIFY_BATCH=1 (development): every job once more as a single alignment -- a batch has to give the same bits
  if (st == MH_OK && n_jobs && getenv("MH_DEBUG_VERIFY_BATCH") != nullptr && !pairs_block) {
    for (size_t i = 0; i < n_jobs; i++) {
      mh_icp_result r2;
      const mh_icp_params* q = params_per_job ? &params[i] : params;
      if (mh_icp_align(maps[i], scans[i], q, T_guesses + 12 * i, priors ? priors[i] : nullptr, &r2, nullptr, nullptr, MH_MEM_HOST) != MH_OK) continue;
      if (memcmp(r2.T, results[i].T, sizeof(r2.T)) != 0 || r2.n_iterations != results[i].n_iterations) {
        double md = 0;
        for (int k = 0; k < 12; k++) md = fmax(md, fabs(r2.T[k] - results[i].T[k]));
        fprintf(stderr, "[MH_DEBUG_VERIFY_BATCH] job %zu of %zu: n = %zu points, batch %u iterations (reason %u, %u pairs) vs single %u (reason %u, %u pairs), max |dT| %.3e; sizes:",
                i, n_jobs, (size_t)scans[i]->n, results[i].n_iterations, results[i].termination_reason, results[i].n_final_pairs, r2.n_iterations,
                r2.termination_reason, r2.n_final_pairs, md);
        for (size_t k = 0; k < n_jobs; k++) fprintf(stderr, " %zu", (size_t)scans[k]->n);
        fprintf(stderr, "\n");
        // which of the two is unstable?  the batch once more, the single once more
        std::vector<mh_icp_result> again(n_jobs);
        if (align_batch_run(n_jobs, maps, scans, params, params_per_job, T_guesses, priors, again.data(), nullptr, pairs_mem) == MH_OK) {
          mh_icp_result r3;
          (void)mh_icp_align(maps[i], scans[i], q, T_guesses + 12 * i, priors ? priors[i] : nullptr, &r3, nullptr, nullptr, MH_MEM_HOST);
          fprintf(stderr, "[MH_DEBUG_VERIFY_BATCH]    second batch == first batch: %d, second batch == single: %d, second single == first single: %d (iterations %u / %u / %u / %u)\n",
                  memcmp(again[i].T, results[i].T, sizeof(r2.T)) == 0, memcmp(again[i].T, r2.T, sizeof(r2.T)) == 0,
                  memcmp(r3.T, r2.T, sizeof(r2.T)) == 0, results[i].n_iterations, again[i].n_iterations, r2.n_iterations, r3.n_iterations);
        }
      }
    }
  }
  return st;
}

static mh_status align_batch_run(size_t n_jobs, const mh_map* const* maps, const mh_scan* const* scans,
                                 const mh_icp_params* params, int32_t params_per_job, const double* T_guesses,
                                 const mh_prior* const* priors, mh_icp_result* results, void* pairs_block, int32_t pairs_mem) {
  MH_REQUIRE(n_jobs == 0 || (maps && scans && params && T_guesses && results), "null argument");
  MH_REQUIRE(!pairs_block || pairs_mem == MH_MEM_HOST || pairs_mem == MH_MEM_DEVICE || pairs_mem == MH_MEM_HOST_PINNED,
             "bad mem space");
  if (n_jobs == 0) return MH_OK;
  auto P = [&](size_t i) { return params_per_job ? &params[i] : params; };
  std::vector<AlignJob> jobs(n_jobs);
  for (size_t i = 0; i < n_jobs; i++) {
    MH_TRY(check_align_args(maps[i], scans[i], P(i), T_guesses + 12 * i, &results[i]));
    for (size_t j = 0; j < i; j++)
      MH_REQUIRE(scans[j]->ctx != scans[i]->ctx, "each job of a batch needs its own context");
    MH_REQUIRE(!pairs_block || scans[i]->ctx->device == scans[0]->ctx->device, "a pairs block needs all jobs on one device");
    jobs[i].defer_upload = n_jobs >= 2;  // a lock-step group uploads its jobs' blocks in one staged copy
    MH_TRY(jobs[i].start(maps[i], scans[i], P(i), T_guesses + 12 * i, priors ? priors[i] : nullptr, &results[i],
                         nullptr, i));
  }
  mh_ctx* lead0 = scans[0]->ctx;
  PairsPlan pp;
  MH_TRY(set_device(lead0));
  MH_TRY(plan_pairs(lead0, jobs, pairs_block, pairs_mem, pp));
  // Lock-step mode: every kernel of an iteration is ONE launch over all jobs of a group (blockIdx.y = job).  The jobs'
  // tails fill each other's idle lanes, which concurrent streams do not achieve (HIP maps them onto four hardware queues
  // whose kernels mostly run one after the other).  A group = the jobs that run the same kernel chain:
  //   quad / tile matcher + k_accum + k_solve (large layers), row matcher with the fused first accumulation (2-12 k
  //   points), row matcher + one-workgroup accumulate-and-solve (<= 2 k points: what lidar3d-default.yaml feeds), and the
  //   same with Matcher_Point2Plane riding along (lidar3d-ndt.yaml);
  // each job keeps its own parameters (iteration budget, schedules, hook check point, prior), state block, termination
  // flag and iteration count.  Jobs whose chain has no lock-step form (or that are alone in their group) take the
  // per-stream path below.
  enum Kind { K_NONE = 0, K_QUAD, K_TILE, K_WAVE, K_ORD, K_ROWF, K_STEP, K_STEP_PL, K_FLAT };
  const bool no_lockstep = getenv("MH_NO_LOCKSTEP") != nullptr;
  const bool batch_prof = !jobs.empty() && jobs[0].prof && !no_lockstep;  // (profile == 2 times job 0's share of a match kernel)
  auto kind_of = [&](const AlignJob& j) -> int {
    if (j.finished || no_lockstep || j.trace || j.prof) return K_NONE;
    // row-kernel layers up to 8 k points: k_step16_b, the chain of a single alignment with the jobs' workgroups side by side -- the
    // same sums in the same order, hence the same bits.  (Round 4 also kept the one-workgroup accumulate-and-solve of round 3 for
    // batches, re-ordered to give those bits: 4 / 8 / 16 sequences 3690 / 4920 / 5770 scans/s against 4092 / 5173 / 5510 this
    // way, NDT pipeline 3966 / 4509 / 3939 against 4780 / 5500 / 5600: removed.)
    if (j.use_step_chain() && !batch_prof) return j.pl ? K_STEP_PL : K_STEP;
    if (j.variant == 4 && !j.pl) return K_QUAD;
    if (j.variant == 9 && !j.pl) return K_FLAT;
    if (j.variant == 6 && !j.pl) return K_TILE;
    if (j.variant == 7 && !j.pl) return K_WAVE;
    if (j.variant == 8 && !j.pl) return K_ORD;
    if (j.variant == 5 && j.fused16 && !j.pl) return K_ROWF;
    return K_NONE;
  };
  struct Group {
    int kind = K_NONE;
    std::vector<AlignJob*> jobs;
    std::vector<size_t> index;
    mh_ctx* lead = nullptr;
    IcpDeviceState* h_states = nullptr;
    const BatchJob* dj = nullptr;
    uint32_t gx_match = 1, gx_acc = 1, gx_cov = 1, gx_step = 1, enq = 0, prof_n = 0, max_iterations = 0, inner = 1, chunk = 10;
    uint32_t src = 2, par = 0, launches = 0;  // k_step16_b: the state block (2 = canonical) and the partials half the next launch reads; launches so far
    bool cov = false, done = false, auto_chunk = false;
    bool loop_wave = false;  // ... as k_icpw_b (point layers): every job its own workgroups
    uint32_t gx_loopw = 1;   // ... whose launch has this many workgroups per job
    bool loop_now = false;   // k_icp16_b: the group's whole loops in ONE launch (decided below; cleared when a job's workgroups gave up)
    uint32_t loop_wgs = 0;   // ... and what it holds of the device's admission count meanwhile
    bool step_chain() const { return kind == K_STEP || kind == K_STEP_PL; }
    bool with_planes() const { return kind == K_STEP_PL; }
  };
  std::vector<Group> groups;
  struct ReleaseLoops {  // whatever way this call ends, the groups' share of the device's loop admission count is given back
    std::vector<Group>* gs;
    ~ReleaseLoops() {
      for (Group& g : *gs)
        if (g.loop_wgs && g.lead) {
          AlignJob::loop_count(g.lead->device).fetch_sub(g.loop_wgs);
          g.loop_wgs = 0;
        }
    }
  } release_loops{&groups};
  const bool want_prof = !jobs.empty() && jobs[0].prof && !no_lockstep;  // profile == 2: the share of job 0's group
  if (want_prof) jobs[0].prof = false;
  for (size_t i = 0; i < n_jobs; i++) {
    const int k = kind_of(jobs[i]);
    if (k == K_NONE) continue;
    const mh_icp_params* q = jobs[i].p;
    Group* g = nullptr;
    for (Group& c : groups)  // same chain, same device, same loop shape
      if (c.kind == k && c.lead->device == jobs[i].ctx->device && c.inner == q->gn.max_inner_iterations &&
          c.cov == (q->compute_covariance != 0))
        g = &c;
    if (!g) {
      groups.emplace_back();
      g = &groups.back();
      g->kind = k;
      g->lead = jobs[i].ctx;
      g->inner = q->gn.max_inner_iterations;
      g->cov = q->compute_covariance != 0;
      g->chunk = q->poll_every ? q->poll_every : 0;
      g->auto_chunk = q->poll_every == 0;
    }
    // automatic chunks: the group's first chunk is as long as its slowest job expects to run (every job's own estimate:
    // AlignJob::start) -- jobs that finish earlier leave their blocks at once, so only what lies beyond the LAST job's end is
    // wasted, while every poll in between drains the device for a host round trip (measured with fixed chunks of 10 on 8
    // sequences: 3.1 polls per alignment)
    if (g->auto_chunk && jobs[i].chunk > g->chunk) g->chunk = jobs[i].chunk;
    g->jobs.push_back(&jobs[i]);
    g->index.push_back(i);
    g->max_iterations = q->max_iterations > g->max_iterations ? q->max_iterations : g->max_iterations;
  }
  {  // a job alone in its group gains nothing from lock step; MH_LOCKSTEP_GROUPS splits the groups further (measured: two
     // groups of the C2 batch overlap one's match launch with the other's short launches for +3 %; default off)
    uint32_t split = 1;
    if (const char* e = getenv("MH_LOCKSTEP_GROUPS")) split = (uint32_t)atoi(e) > 0 ? (uint32_t)atoi(e) : 1u;
    std::vector<Group> kept;
    for (Group& g : groups) {
      if (g.jobs.size() < 2) continue;
      uint32_t parts = split;
      if (parts > g.jobs.size() / 2) parts = (uint32_t)(g.jobs.size() / 2);
      if (parts < 1 || pp.want) parts = 1;  // (the pairs block is compacted by one launch over one group's jobs)
      for (uint32_t part = 0; part < parts; part++) {
        Group h = g;
        h.jobs.clear();
        h.index.clear();
        for (size_t a = 0; a < g.jobs.size(); a++)
          if (a * parts / g.jobs.size() == part) {
            h.jobs.push_back(g.jobs[a]);
            h.index.push_back(g.index[a]);
          }
        h.lead = h.jobs[0]->ctx;
        kept.push_back(h);
      }
    }
    groups.swap(kept);
    if (groups.size() > 64) groups.resize(64);  // (their jobs fall through to the per-stream path)
  }
  std::vector<char> in_group(n_jobs, 0);
  for (Group& g : groups)
    for (size_t i : g.index) in_group[i] = 1;
  if (want_prof && !in_group[0]) jobs[0].prof = true;  // job 0 goes the per-stream way: its own events
  bool pairs_by_group = pp.want && groups.size() == 1 && groups[0].jobs.size() == (size_t)std::count_if(jobs.begin(), jobs.end(), [](const AlignJob& j) { return !j.finished; });

  if (!groups.empty()) {
    constexpr size_t kBlockBytes = kParamsOffset + sizeof(IcpDeviceParams);  // one job's [state | params] block
    static_assert(kBlockBytes % 4 == 0 && sizeof(BatchJob) % 8 == 0, "staging layout");
    for (Group& g : groups) {
      const uint32_t A = (uint32_t)g.jobs.size();
      mh_ctx* lead = g.lead;
      MH_TRY(set_device(lead));
      hipStream_t s = lead->stream;
#ifdef MH_DEV_VARIANTS
      if (g.kind == K_TILE || g.kind == K_WAVE)
        for (AlignJob* j : g.jobs) MH_TRY(scan_tiles_ready(j->scan));  // tile counts (the builds were queued by start())
#endif
      MH_TRY(order_after_job_streams(lead, g.jobs));
      size_t stage_bytes = 0;
      for (AlignJob* j : g.jobs) stage_bytes += kBlockBytes + j->nsched_pending * sizeof(double);
      const size_t need = A * sizeof(IcpDeviceState) + A * sizeof(BatchJob) + stage_bytes;
      MH_TRY(lead->batch_desc.reserve(A * sizeof(BatchJob) + stage_bytes));  // descriptors | staged blocks
      MH_TRY(lead->batch_states.reserve(A * sizeof(IcpDeviceState)));
      if (lead->h_batch_cap < need) {
        if (lead->h_batch) (void)hipHostFree(lead->h_batch);
        lead->h_batch = nullptr;
        lead->h_batch_cap = 0;
        MH_HIP(hipHostMalloc(&lead->h_batch, need, hipHostMallocDefault));
        lead->h_batch_cap = need;
      }
      g.h_states = reinterpret_cast<IcpDeviceState*>(lead->h_batch);
      BatchJob* h_desc = reinterpret_cast<BatchJob*>(reinterpret_cast<char*>(lead->h_batch) + A * sizeof(IcpDeviceState));
      char* h_stage = reinterpret_cast<char*>(h_desc) + A * sizeof(BatchJob);
      size_t off = 0;
      for (uint32_t a = 0; a < A; a++) {
        AlignJob& j = *g.jobs[a];
        BatchJob& d = h_desc[a];
        fill_batch_desc(j, d);
        if (pairs_by_group) set_pairs_fields(pp, g.index[a], d);
        d.stage_off = (uint32_t)(off / 4);
        uint32_t bm = (uint32_t)((4ull * d.n + kBlock - 1) / kBlock);  // quad
        if (g.kind == K_ROWF) bm = d.nbm;
        if (g.kind == K_FLAT) bm = nblk_flat(d.n);
        if (g.kind == K_TILE) bm = d.n_tiles;
        if (g.kind == K_WAVE) bm = d.n_tiles;
        if (g.step_chain()) {  // all jobs' workgroups resident at once: kStepMaxWorkgroups shared between them
          static const uint32_t cap_env = getenv("MH_STEP_WGS") ? (uint32_t)std::max(1, atoi(getenv("MH_STEP_WGS"))) : 0u;  // (development)
          const uint32_t cap = cap_env ? cap_env : (kStepMaxWorkgroups / A ? kStepMaxWorkgroups / A : 1u);
          const uint32_t ng = (d.n + kStepPoints - 1) / kStepPoints;
          const uint32_t nw = ng < cap ? ng : cap;
          g.gx_step = nw > g.gx_step ? nw : g.gx_step;
        }
        g.gx_match = bm > g.gx_match ? bm : g.gx_match;
        g.gx_acc = d.nba > g.gx_acc ? d.nba : g.gx_acc;
        g.gx_cov = d.nb > g.gx_cov ? d.nb : g.gx_cov;
        // this job's [state | params] mirror and its schedules into the staging area
        memcpy(h_stage + off, j.ctx->h_state, kBlockBytes);
        memcpy(h_stage + off + kBlockBytes, j.ctx->h_sched, j.nsched_pending * sizeof(double));
        off += kBlockBytes + j.nsched_pending * sizeof(double);
        j.defer_upload = false;
      }
      // Small layers: the whole loops of the group's jobs in ONE launch (k_icp16_b) when every job's layer has at most kLoopMaxGroups
      // groups, a workgroup takes at most kLoopGroupsPerWg of them, and the launch's workgroups are admitted (all resident together,
      // beside the one-launch loops of single alignments running on the device).  MH_NO_LOOP16 / MH_NO_LOOP16_BATCH: the chain.
      if (g.step_chain() && getenv("MH_NO_LOOP16") == nullptr && getenv("MH_NO_LOOP16_BATCH") == nullptr &&
          !AlignJob::loop_holdoff(lead->device, 0)) {
        bool fits = true;
        // k_icpw_b (point layers): every job its own workgroups of 128 points; k_icp16_b (NDT maps, MH_NO_LOOPW): the jobs share
        // kStepMaxWorkgroups workgroups of 32 points, a workgroup taking several groups
        g.loop_wave = !g.with_planes() && loop_wave_enabled(true);
        uint32_t units = 0, gx_w = 1;
        for (uint32_t a = 0; a < A; a++) {
          const uint32_t ng = (h_desc[a].n + kStepPoints - 1) / kStepPoints;
          const uint32_t nw = ng < g.gx_step ? ng : g.gx_step;
          if (g.loop_wave) {
            fits = fits && ng <= kLwMaxGroups && g.jobs[a]->sk.max_iterations > 0;
            const uint32_t w = (ng + kLwGroups - 1) / kLwGroups;
            units += w;
            gx_w = w > gx_w ? w : gx_w;
          } else {
            fits = fits && ng <= kLoopMaxGroups && (nw == 0 || (ng + nw - 1) / nw <= kLoopGroupsPerWg) && g.jobs[a]->sk.max_iterations > 0;
          }
        }
        if (!g.loop_wave) units = kLoopUnitsPerCu * g.gx_step * A;
        if (fits && AlignJob::loop_admit(lead->device, units)) {
          g.loop_now = true;
          g.loop_wgs = units;
          g.gx_loopw = gx_w;
          for (uint32_t a = 0; a < A && g.loop_now; a++) {
            AlignJob& j = *g.jobs[a];
            if (g.loop_wave && (map_ensure_qidx(j.map, s) != MH_OK || !j.map->view().pts_q)) {  // (s waits for every job's stream: above)
              g.loop_now = false;
              break;
            }
            if (g.loop_wave) h_desc[a].map = j.map->view();  // (with the sub-voxel index)
            if (j.ctx->loop_x.bytes < kLoopExchangeBytes) {
              if (j.ctx->loop_x.reserve(kLoopExchangeBytes) != MH_OK) {
                g.loop_now = false;
                break;
              }
              (void)hipMemsetAsync(j.ctx->loop_x.p, 0, kLoopExchangeBytes, s);
            }
            const uint32_t max_steps = j.p->max_iterations * j.p->gn.max_inner_iterations + 1u;
            h_desc[a].loop_xa = j.ctx->loop_x.p;
            h_desc[a].loop_xb = static_cast<char*>(j.ctx->loop_x.p) + 2 * (size_t)kAccN * kLoopRowStride * 16;
            h_desc[a].loop_serial0 = j.ctx->loop_serial;
            h_desc[a].loop_pad = getenv("MH_LOOP16_TEST_ABANDON") ? 1u : 0u;
            j.ctx->loop_serial += max_steps + 2u;
          }
          if (!g.loop_now) {
            AlignJob::loop_count(lead->device).fetch_sub(g.loop_wgs);
            g.loop_wgs = 0;
          }
        }
      }
      // descriptors and staged blocks in ONE copy, then a scatter kernel writes every job's block where it lives
      MH_HIP(hipMemcpyAsync(lead->batch_desc.p, h_desc, A * sizeof(BatchJob) + stage_bytes, hipMemcpyHostToDevice, s));
      g.dj = lead->batch_desc.as<BatchJob>();
      hipLaunchKernelGGL(k_scatter_blocks, dim3(A), dim3(256), 0, s, g.dj,
                         reinterpret_cast<const uint32_t*>(lead->batch_desc.as<char>() + A * sizeof(BatchJob)),
                         (uint32_t)(kBlockBytes / 4));
    }
    // (Round 4 also built the streaming control of AlignJob::run_streaming for a whole lock-step group -- every job publishing in
    // its own progress word, the host following the slowest job still running -- and measured it on 4 / 8 / 16 sequences in one
    // process: 4073 / 4904 / 6597 scans/s against 4040 / 5255 / 6560 with the chunks below, NDT pipeline 4475 / 4963 / 5120
    // against 4186 / 5120 / 5245.  The spinning leader thread takes a core from the seven threads that queue uploads, filters
    // and map updates beside it, and a group's tail is amortised over its jobs anyway.  Removed; single alignments keep it.)
    for (;;) {
      bool any = false;
      uint32_t m_of[64] = {0};
      // enqueue one chunk per unfinished group, iteration by iteration across the groups so that their launches interleave
      uint32_t m_max = 0;
      for (size_t gi = 0; gi < groups.size(); gi++) {
        Group& g = groups[gi];
        if (g.done) continue;
        any = true;
        m_of[gi] = (g.max_iterations - g.enq) < g.chunk ? (g.max_iterations - g.enq) : g.chunk;
        if (g.loop_now) {  // everything in one launch, now
          m_of[gi] = g.max_iterations - g.enq;
          const uint32_t A = (uint32_t)g.jobs.size();
          g_loop16_runs.fetch_add(A);
          if (g.loop_wave) hipLaunchKernelGGL(k_icpw_b, dim3(g.gx_loopw, A), dim3(kLwThreads), 0, g.lead->stream, g.dj);
          else if (g.with_planes()) hipLaunchKernelGGL(k_icp16_b<true>, dim3(g.gx_step, A), dim3(kSolveThreads), 0, g.lead->stream, g.dj);
          else hipLaunchKernelGGL(k_icp16_b<false>, dim3(g.gx_step, A), dim3(kSolveThreads), 0, g.lead->stream, g.dj);
          continue;
        }
        m_max = m_of[gi] > m_max ? m_of[gi] : m_max;
      }
      if (!any) break;
      for (uint32_t it = 0; it < m_max; it++)
        for (size_t gi = 0; gi < groups.size(); gi++) {
          Group& g = groups[gi];
          if (g.done || g.loop_now || it >= m_of[gi]) continue;
          hipStream_t s = g.lead->stream;
          const uint32_t A = (uint32_t)g.jobs.size();
          const bool pr = want_prof && gi == 0 && g.jobs[0] == &jobs[0];
          if (g.step_chain()) {
            for (uint32_t in = 0; in < g.inner; in++) {
              if (g.with_planes()) hipLaunchKernelGGL(k_step16_b<true>, dim3(g.gx_step, A), dim3(kSolveThreads), 0, s, g.dj, g.src, g.par, 0u, g.launches);
              else hipLaunchKernelGGL(k_step16_b<false>, dim3(g.gx_step, A), dim3(kSolveThreads), 0, s, g.dj, g.src, g.par, 0u, g.launches);
              g.src = g.src == 2u ? 0u : (g.src ^ 1u);
              g.par ^= 1u;
              g.launches++;
            }
            continue;
          }
          if (pr) MH_HIP(hipEventRecord(g.lead->prof_ev[2 * g.prof_n], s));
          switch (g.kind) {
            case K_ROWF: hipLaunchKernelGGL(k_match16f_b, dim3(g.gx_match, A), dim3(kBlock), 0, s, g.dj); break;
#ifdef MH_DEV_VARIANTS
            case K_TILE: hipLaunchKernelGGL(k_match_tile_b, dim3(g.gx_match, A), dim3(kTileThreads), 0, s, g.dj); break;
            case K_WAVE:
              if (wave_lds_env()) hipLaunchKernelGGL(k_match_wave_dense_b<true>, dim3(g.gx_match, A), dim3(64), 0, s, g.dj);
              else hipLaunchKernelGGL(k_match_wave_dense_b<false>, dim3(g.gx_match, A), dim3(64), 0, s, g.dj);
              hipLaunchKernelGGL(k_match_wave_sparse_b, dim3(g.gx_match, A), dim3(kBlock), 0, s, g.dj);
              break;
            case K_ORD: hipLaunchKernelGGL(k_match4o_b, dim3(g.gx_match, A), dim3(kBlock), 0, s, g.dj); break;
#endif
            case K_FLAT: hipLaunchKernelGGL(k_match_flat_b, dim3(g.gx_match, A), dim3(kFlatThreads), 0, s, g.dj); break;
            default: hipLaunchKernelGGL(k_match4_b, dim3(g.gx_match, A), dim3(kBlock), 0, s, g.dj); break;
          }
          if (pr) {
            MH_HIP(hipEventRecord(g.lead->prof_ev[2 * g.prof_n + 1], s));
            g.prof_n++;
          }
          if (g.kind != K_ROWF) hipLaunchKernelGGL(g.kind == K_FLAT ? k_accum_b<true> : k_accum_b<false>, dim3(g.gx_acc, A), dim3(kBlock), 0, s, g.dj, 1u);
          hipLaunchKernelGGL(k_solve_b, dim3(1, A), dim3(kSolveThreads), 0, s, g.dj, 1u);
          for (uint32_t in = 1; in < g.inner; in++) {
            hipLaunchKernelGGL(g.kind == K_FLAT ? k_accum_b<true> : k_accum_b<false>, dim3(g.gx_acc, A), dim3(kBlock), 0, s, g.dj, 0u);
            hipLaunchKernelGGL(k_solve_b, dim3(1, A), dim3(kSolveThreads), 0, s, g.dj, 0u);
          }
        }
      for (size_t gi = 0; gi < groups.size(); gi++) {
        Group& g = groups[gi];
        if (g.done) continue;
        hipStream_t s = g.lead->stream;
        const uint32_t A = (uint32_t)g.jobs.size();
        if (g.step_chain() && !g.loop_now) {  // the pending Gauss-Newton step of every job, into the canonical state blocks
          if (g.with_planes()) hipLaunchKernelGGL(k_step16_b<true>, dim3(1, A), dim3(kSolveThreads), 0, s, g.dj, g.src, g.par, 1u, g.launches);
          else hipLaunchKernelGGL(k_step16_b<false>, dim3(1, A), dim3(kSolveThreads), 0, s, g.dj, g.src, g.par, 1u, g.launches);
          g.src = 2;
          g.launches++;
        }
        if (g.cov) {  // no-ops for jobs whose loop has not terminated
          hipLaunchKernelGGL(k_cov_prepare_b, dim3(1, A), dim3(64), 0, s, g.dj);
          hipLaunchKernelGGL(k_cov_accum_b, dim3(g.gx_cov, A), dim3(kBlock), 0, s, g.dj);
          if (g.with_planes()) hipLaunchKernelGGL(k_cov_accum_plbuf_b, dim3(g.gx_cov, A), dim3(kBlock), 0, s, g.dj);
          hipLaunchKernelGGL(k_cov_finalize_b, dim3(1, A), dim3(kSolveThreads), 0, s, g.dj);
        }
        hipLaunchKernelGGL(k_gather_states, dim3(A), dim3(256), 0, s, g.dj, g.lead->batch_states.as<IcpDeviceState>());
        MH_HIP(hipGetLastError());
        MH_HIP(hipMemcpyAsync(g.h_states, g.lead->batch_states.p, A * sizeof(IcpDeviceState), hipMemcpyDeviceToHost, s));
      }
      for (size_t gi = 0; gi < groups.size(); gi++) {
        Group& g = groups[gi];
        if (g.done) continue;
        const hipError_t we = mh::wait_stream(g.lead->stream);
        if (g.loop_wgs) {
          AlignJob::loop_count(g.lead->device).fetch_sub(g.loop_wgs);
          g.loop_wgs = 0;
        }
        MH_HIP(we);
        if (g.loop_now) {
          // a job whose workgroups gave up waiting for each other left done == 0 and its canonical state block as uploaded: the
          // group goes on launch by launch (k_step16_b from the start; the jobs that did finish are no-ops there)
          g.loop_now = false;
          bool abandoned = false;
          for (size_t a = 0; a < g.jobs.size(); a++) {
            const IcpDeviceState& h = g.h_states[a];
            if (g.jobs[a]->finished || (h.done && !h.handover_timeouts)) continue;
            abandoned = true;
            g_loop16_fallbacks.fetch_add(1);
            MH_HIP(hipMemsetAsync(&g.jobs[a]->ctx->d_state->handover_timeouts, 0, sizeof(uint32_t) * 10, g.lead->stream));
          }
          if (!abandoned) AlignJob::loop_holdoff(g.lead->device, -1);  // (a clean run ends a streak of abandoned loops)
          if (abandoned) {
            if (getenv("MH_LOOP16_TEST_ABANDON") == nullptr) AlignJob::loop_holdoff(g.lead->device, 1);
            for (size_t a = 0; a < g.jobs.size(); a++) {
              AlignJob& j = *g.jobs[a];
              const IcpDeviceState& h = g.h_states[a];
              if (j.finished || !(h.done && !h.handover_timeouts)) continue;
              memcpy(j.ctx->h_state, &h, sizeof(IcpDeviceState));
              j.enqueued = j.p->max_iterations;
              MH_TRY(j.poll(true));
            }
            g.enq = 0;
            g.src = 2;
            g.par = 0;
            g.launches = 0;
            continue;  // (not done: the next round enqueues the chain's first chunk)
          }
        }
        g.enq += m_of[gi];
        g.done = true;
        if (g.auto_chunk) g.chunk = 8;  // follow-up chunks
        for (size_t a = 0; a < g.jobs.size(); a++) {
          AlignJob& j = *g.jobs[a];
          if (j.finished) continue;
          memcpy(j.ctx->h_state, &g.h_states[a], sizeof(IcpDeviceState));
          j.enqueued = g.enq < j.p->max_iterations ? g.enq : j.p->max_iterations;
          MH_TRY(j.poll(true));
          g.done = g.done && j.finished;
        }
      }
    }
    if (want_prof && groups[0].jobs[0] == &jobs[0]) {  // the match step of job 0 = its share of its group's lock-step launches
      Group& g = groups[0];
      float ms = 0.f;
      double sum = 0.0;
      const mh_icp_result* r0 = jobs[0].res;
      uint32_t live = r0->n_iterations + ((r0->termination_reason == MH_TERM_MAX_ITERATIONS) ? 0u : 1u);
      if (live > g.prof_n) live = g.prof_n;
      for (uint32_t i = 0; i < live; i++) {
        MH_HIP(hipEventElapsedTime(&ms, g.lead->prof_ev[2 * i], g.lead->prof_ev[2 * i + 1]));
        sum += ms;
      }
      jobs[0].res->n_match_launches = live;
      jobs[0].res->match_kernel_ms = sum / (double)g.jobs.size();
      jobs[0].res->total_ms = 0.0;
    }
    if (pairs_by_group) {
      MH_TRY(finish_pairs(lead0, groups[0].lead->stream, pp, groups[0].dj, (uint32_t)groups[0].jobs.size(), groups[0].gx_cov));
      return MH_OK;
    }
  }
  // everything that is not in a lock-step group: one stream per job, chunks enqueued round robin
  for (size_t i = 0; i < n_jobs; i++)
    if (!in_group[i]) MH_TRY(jobs[i].flush_deferred());
  for (;;) {
    bool any = false;
    for (size_t i = 0; i < n_jobs; i++)
      if (!in_group[i] && !jobs[i].finished) {
        MH_TRY(jobs[i].enqueue_chunk());
        any = true;
      }
    if (!any) break;
    for (size_t i = 0; i < n_jobs; i++)
      if (!in_group[i]) MH_TRY(jobs[i].poll());
  }
  if (pp.want) {
    // every job has terminated and its stream is drained: one compaction launch over all of them on the first job's stream
    std::vector<size_t> act;
    for (size_t i = 0; i < n_jobs; i++)
      if (!jobs[i].trivial) act.push_back(i);
    if (act.empty()) return MH_OK;
    mh_ctx* lead = lead0;
    MH_TRY(set_device(lead));
    const uint32_t A = (uint32_t)act.size();
    MH_TRY(lead->batch_desc.reserve(A * sizeof(BatchJob)));
    if (lead->h_batch_cap < A * sizeof(BatchJob)) {
      if (lead->h_batch) (void)hipHostFree(lead->h_batch);
      lead->h_batch = nullptr;
      lead->h_batch_cap = 0;
      MH_HIP(hipHostMalloc(&lead->h_batch, A * sizeof(BatchJob), hipHostMallocDefault));
      lead->h_batch_cap = A * sizeof(BatchJob);
    }
    BatchJob* h_desc = reinterpret_cast<BatchJob*>(lead->h_batch);
    uint32_t gx_cov = 1;
    for (uint32_t a = 0; a < A; a++) {
      fill_batch_desc(jobs[act[a]], h_desc[a]);
      set_pairs_fields(pp, act[a], h_desc[a]);
      gx_cov = h_desc[a].nb > gx_cov ? h_desc[a].nb : gx_cov;
    }
    MH_HIP(hipMemcpyAsync(lead->batch_desc.p, h_desc, A * sizeof(BatchJob), hipMemcpyHostToDevice, lead->stream));
    MH_TRY(finish_pairs(lead, lead->stream, pp, lead->batch_desc.as<BatchJob>(), A, gx_cov));
    if (pp.mem != MH_MEM_HOST) MH_HIP(mh::wait_stream(lead->stream));  // h_batch is reused by the next batch
  }
  return MH_OK;
}
