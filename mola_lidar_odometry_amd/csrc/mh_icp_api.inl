// mh_icp_api.inl -- the matcher- and solver-granular entry points (mh_nn_search*, mh_gn_solve, mh_covariance: what the
// Matcher_* / Solver_GaussNewton plugin classes call).  Included by mh_icp.hip inside its extern "C" block.

namespace {
#ifdef MH_DEV_VARIANTS
// MH_MATCH=t: the matcher-granular entry points run the tile matcher too (the parity tests drive every search kernel
// through mh_nn_search / mh_nn_search_dense); thr2 = +inf: no threshold
mh_status launch_tile_search(const mh_map* map, const mh_scan* scan, const double T[12], float thr2, float ang2) {
  mh_ctx* ctx = scan->ctx;
  const bool wave = tile_points_for_env() == 64u;
  MH_TRY(scan_build_tiles(scan, map->inv_vs, wave ? 64u : 256u));
  MH_TRY(scan_tiles_ready(scan));
  MH_TRY(map_ensure_qidx(map, ctx->stream));  // (sparse tiles are searched by quads)
  MH_HIP(mh::wait_stream(ctx->stream));  // the pinned state mirror may still be travelling
  init_state(ctx->h_state, T);
  ctx->h_state->cur_thr2 = thr2;
  ctx->h_state->cur_ang2 = ang2;
  MH_HIP(hipMemcpyAsync(ctx->d_state, ctx->h_state, sizeof(IcpDeviceState), hipMemcpyHostToDevice, ctx->stream));
  if (scan->n_tiles && wave)
    MH_LAUNCH_WAVE(ctx->stream, ctx->d_state, scan, map->view(), ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(), MH_WT_NULL);
  else if (scan->n_tiles)
    hipLaunchKernelGGL(k_match_tile, dim3(scan->n_tiles), dim3(kTileThreads), 0, ctx->stream, ctx->d_state, scan->sx, scan->sy,
                       scan->sz, scan->perm, scan->tile_start, scan->n_tiles, map->view(), ctx->pair_q.as<float4>(),
                       ctx->pair_gidx.as<uint32_t>()
#ifdef MH_DEBUG_WAVETRACE
                       , (unsigned long long*)nullptr
#endif
    );
  MH_HIP(hipGetLastError());
  return MH_OK;
}
inline bool tile_search_forced() {
  const char* e = getenv("MH_MATCH");
  return e && (e[0] == 't' || e[0] == 'w');
}
#else  // the shipped library: the matcher-granular entry points run k_match<false, 1> whatever MH_MATCH says
inline bool tile_search_forced() { return false; }
inline mh_status launch_tile_search(const mh_map*, const mh_scan*, const double*, float, float) { return MH_OK; }
#endif
}  // namespace

mh_status mh_nn_search(const mh_map* map, const mh_scan* scan, const double T[12], double threshold,
                       double threshold_angular_deg, const mh_pairs_out* out, int32_t mem, mh_match_info* info) {
  MH_REQUIRE(map && scan && T, "null argument");
  MH_REQUIRE(mem == MH_MEM_HOST || mem == MH_MEM_DEVICE, "bad mem space");
  MH_REQUIRE(map->ctx->device == scan->ctx->device, "map and scan live on different devices");
  MH_REQUIRE(pose_ok(T), "non-finite pose");
  mh_ctx* ctx = scan->ctx;
  MH_TRY(set_device(ctx));
  MH_TRY(map_ready_on(map, ctx->stream));
  if (info) {
    info->n_pairs = 0;
    info->potential_pairings = scan->n;  // counted before any test (App.B U6)
  }
  if (scan->n == 0) return MH_OK;
  MH_TRY(ensure_state(ctx));
  MH_TRY(ensure_pair_buffers(ctx, scan->n));
  PoseArg Ta;
  for (int i = 0; i < 12; i++) Ta.m[i] = T[i];
  MatchK mk{};
  SolveK sk0{};
  const double ang = threshold_angular_deg * 3.14159265358979323846 / 180.0;
  mk.ang2 = (float)(ang * ang);
  MH_TRY(upload_params(ctx, mk, sk0));
  if (tile_search_forced())
    MH_TRY(launch_tile_search(map, scan, T, (float)(threshold * threshold), mk.ang2));
  else
    hipLaunchKernelGGL((k_match<false, 1>), dim3(nblk(scan->n)), dim3(kBlock), 0, ctx->stream, ctx->d_state, Ta,
                       (float)(threshold * threshold), 1u, &ctx->d_params->mk, scan->x, scan->y, scan->z, (uint32_t)scan->n, map->view(),
                       ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(), (double*)nullptr, 0u);
  MH_HIP(hipGetLastError());
  mh_pairs_out none{};
  uint64_t np = 0;
  MH_TRY(compact_pairs(ctx, scan->n, out ? out : &none, mem, &np));
  if (info) info->n_pairs = np;
  return MH_OK;
}

mh_status mh_nn_search_k(const mh_map* map, const mh_scan* scan, const double T[12], double threshold,
                         double threshold_angular_deg, uint32_t pairings_per_point, const mh_pairs_out* out, int32_t mem,
                         mh_match_info* info) {
  MH_REQUIRE(map && scan && T, "null argument");
  MH_REQUIRE(mem == MH_MEM_HOST || mem == MH_MEM_DEVICE, "bad mem space");
  MH_REQUIRE(pairings_per_point >= 1 && pairings_per_point <= (uint32_t)kMaxKnn, "pairings_per_point must be 1..MH_MAX_PAIRINGS_PER_POINT");
  MH_REQUIRE(map->ctx->device == scan->ctx->device, "map and scan live on different devices");
  MH_REQUIRE(pose_ok(T), "non-finite pose");
  MH_REQUIRE((uint64_t)scan->n * pairings_per_point < 0xFFFFFFFFull, "scan size * pairings_per_point does not fit 32 bits");
  if (pairings_per_point == 1) return mh_nn_search(map, scan, T, threshold, threshold_angular_deg, out, mem, info);
  mh_ctx* ctx = scan->ctx;
  MH_TRY(set_device(ctx));
  MH_TRY(map_ready_on(map, ctx->stream));
  const uint32_t k = pairings_per_point;
  if (info) {
    info->n_pairs = 0;
    info->potential_pairings = (uint64_t)scan->n * k;  // pcLocal.size() * pairingsPerPoint, counted before any test (App.B U6)
  }
  if (scan->n == 0) return MH_OK;
  const size_t nk = scan->n * (size_t)k;
  MH_TRY(ensure_pair_buffers(ctx, nk));
  PoseArg Ta;
  for (int i = 0; i < 12; i++) Ta.m[i] = T[i];
  const double ang = threshold_angular_deg * 3.14159265358979323846 / 180.0;
  hipLaunchKernelGGL(k_match_kbest, dim3(nblk(scan->n)), dim3(kBlock), 0, ctx->stream, Ta, (float)(threshold * threshold),
                     (float)(ang * ang), k, scan->x, scan->y, scan->z, (uint32_t)scan->n, map->view(), ctx->pair_q.as<float4>(),
                     ctx->pair_gidx.as<uint32_t>());
  MH_HIP(hipGetLastError());
  mh_pairs_out none{};
  uint64_t np = 0;
  MH_TRY(compact_pairs(ctx, nk, out ? out : &none, mem, &np));
  // compact_pairs numbers the ENTRIES: entry e belongs to local point e / k
  if (out && out->local_idx && np) {
    if (mem == MH_MEM_HOST) {
      for (uint64_t e = 0; e < np; e++) out->local_idx[e] /= k;
    } else {
      hipLaunchKernelGGL(k_div_idx, dim3(nblk(np)), dim3(kBlock), 0, ctx->stream, out->local_idx, (uint32_t)np, k);
      MH_HIP(hipGetLastError());
      MH_HIP(mh::wait_stream(ctx->stream));
    }
  }
  if (info) info->n_pairs = np;
  return MH_OK;
}

mh_status mh_nn_search_dense(const mh_map* map, const mh_scan* scan, const double T[12], uint32_t* global_idx, float* gx,
                             float* gy, float* gz, float* d2, int32_t mem) {
  MH_REQUIRE(map && scan && T, "null argument");
  MH_REQUIRE(mem == MH_MEM_HOST || mem == MH_MEM_DEVICE, "bad mem space");
  MH_REQUIRE(map->ctx->device == scan->ctx->device, "map and scan live on different devices");
  MH_REQUIRE(pose_ok(T), "non-finite pose");
  mh_ctx* ctx = scan->ctx;
  MH_TRY(set_device(ctx));
  MH_TRY(map_ready_on(map, ctx->stream));
  const size_t n = scan->n;
  if (n == 0) return MH_OK;
  MH_TRY(ensure_state(ctx));
  MH_TRY(ensure_pair_buffers(ctx, n));
  PoseArg Ta;
  for (int i = 0; i < 12; i++) Ta.m[i] = T[i];
  MatchK mk{};
  SolveK sk0{};
  hipStream_t s = ctx->stream;
  MH_TRY(upload_params(ctx, mk, sk0));
  if (tile_search_forced())
    MH_TRY(launch_tile_search(map, scan, T, __builtin_inff(), 0.f));
  else
    hipLaunchKernelGGL((k_match<false, 1>), dim3(nblk(n)), dim3(kBlock), 0, s, ctx->d_state, Ta, 0.f, 0u,
                       &ctx->d_params->mk, scan->x,
                       scan->y, scan->z, (uint32_t)n, map->view(), ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>(),
                       (double*)nullptr, 0u);
  uint32_t* o_gi = global_idx;
  float *o_x = gx, *o_y = gy, *o_z = gz, *o_d2 = d2;
  const size_t n4 = ((n + 63) / 64) * 64;
  if (mem == MH_MEM_HOST) {
    MH_TRY(ctx->compact.reserve(5 * n4 * 4));
    char* st = ctx->compact.as<char>();
    o_gi = global_idx ? (uint32_t*)st : nullptr;
    o_x = gx ? (float*)(st + n4 * 4) : nullptr;
    o_y = gy ? (float*)(st + 2 * n4 * 4) : nullptr;
    o_z = gz ? (float*)(st + 3 * n4 * 4) : nullptr;
    o_d2 = d2 ? (float*)(st + 4 * n4 * 4) : nullptr;
  }
  hipLaunchKernelGGL(k_unpack_dense, dim3(nblk(n)), dim3(kBlock), 0, s, ctx->pair_gidx.as<uint32_t>(),
                     ctx->pair_q.as<float4>(), (uint32_t)n, o_gi, o_x, o_y, o_z, o_d2);
  MH_HIP(hipGetLastError());
  MH_HIP(mh::wait_stream(s));
  if (mem == MH_MEM_HOST) {
    if (global_idx) MH_HIP(hipMemcpy(global_idx, o_gi, n * 4, hipMemcpyDeviceToHost));
    if (gx) MH_HIP(hipMemcpy(gx, o_x, n * 4, hipMemcpyDeviceToHost));
    if (gy) MH_HIP(hipMemcpy(gy, o_y, n * 4, hipMemcpyDeviceToHost));
    if (gz) MH_HIP(hipMemcpy(gz, o_z, n * 4, hipMemcpyDeviceToHost));
    if (d2) MH_HIP(hipMemcpy(d2, o_d2, n * 4, hipMemcpyDeviceToHost));
  }
  return MH_OK;
}

// compaction of the context's point-to-plane pairing buffers into caller arrays
static mh_status compact_pl_pairs(mh_ctx* ctx, size_t n, const mh_pairs_pl_out* out, int32_t mem, uint64_t* n_pairs_out) {
  hipStream_t s = ctx->stream;
  const uint32_t nb = nblk(n);
  const size_t n4 = ((n + 63) / 64) * 64;
  // layout: flags[n4] | counts[nb] | offsets[nb] | total[1] | (host staging) li,cx,cy,cz,nx,ny,nz [n4 each]
  const size_t hdr = ((n4 + (size_t)2 * nb + 1) * 4 + 255) / 256 * 256;
  MH_TRY(ctx->compact.reserve(hdr + 7 * n4 * 4));
  uint32_t* flags = ctx->compact.as<uint32_t>();
  uint32_t* counts = flags + n4;
  uint32_t* offsets = counts + nb;
  uint32_t* total = offsets + nb;
  char* stage = ctx->compact.as<char>() + hdr;
  void* o[7] = {out->local_idx, out->cx, out->cy, out->cz, out->nx, out->ny, out->nz};
  void* d[7];
  for (int a = 0; a < 7; a++) d[a] = (mem == MH_MEM_DEVICE) ? o[a] : (o[a] ? (void*)(stage + (size_t)a * n4 * 4) : nullptr);
  uint32_t h_total = 0;
  if (n) {
    hipLaunchKernelGGL(k_pl_flags, dim3(nb), dim3(kBlock), 0, s, ctx->pl_c.as<float4>(), (uint32_t)n, flags);
    hipLaunchKernelGGL(k_count_valid, dim3(nb), dim3(kBlock), 0, s, flags, (uint32_t)n, counts);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, s, counts, nb, offsets, total);
    hipLaunchKernelGGL(k_compact_pl, dim3(nb), dim3(kBlock), 0, s, flags, ctx->pl_c.as<float4>(), ctx->pl_n.as<float4>(),
                       (uint32_t)n, offsets, (uint32_t*)d[0], (float*)d[1], (float*)d[2], (float*)d[3], (float*)d[4],
                       (float*)d[5], (float*)d[6]);
    MH_HIP(hipGetLastError());
    MH_HIP(hipMemcpyAsync(&h_total, total, 4, hipMemcpyDeviceToHost, s));
    MH_HIP(mh::wait_stream(s));
  }
  if (mem == MH_MEM_HOST && h_total)
    for (int a = 0; a < 7; a++)
      if (o[a]) MH_HIP(hipMemcpy(o[a], d[a], (size_t)h_total * 4, hipMemcpyDeviceToHost));
  if (n_pairs_out) *n_pairs_out = h_total;
  return MH_OK;
}

mh_status mh_nn_search_pt2pl(const mh_map* map, const mh_scan* scan, const double T[12], double distance_threshold,
                             uint32_t mode, const mh_pairs_pl_out* out, int32_t mem, mh_match_info* info) {
  MH_REQUIRE(map && scan && T, "null argument");
  MH_REQUIRE(mode == MH_PT2PL_PLANE_DISTANCE || mode == MH_PT2PL_CENTROID_DISTANCE, "bad pt2pl mode");
  distance_threshold = (mode == MH_PT2PL_CENTROID_DISTANCE ? -1.0 : 1.0) * fabs(distance_threshold);
  MH_REQUIRE(mem == MH_MEM_HOST || mem == MH_MEM_DEVICE, "bad mem space");
  MH_REQUIRE(map->ctx->device == scan->ctx->device, "map and scan live on different devices");
  MH_REQUIRE(pose_ok(T), "non-finite pose");
  MH_REQUIRE(map->view().ndt, "the map carries no NDT statistics (build it with ndt_max_eigen_ratio > 0)");
  mh_ctx* ctx = scan->ctx;
  MH_TRY(set_device(ctx));
  MH_TRY(map_ready_on(map, ctx->stream));
  if (info) {
    info->n_pairs = 0;
    info->potential_pairings = scan->n;
  }
  if (scan->n == 0) return MH_OK;
  MH_TRY(ensure_state(ctx));
  MH_TRY(ensure_pl_buffers(ctx, scan->n));
  PoseArg Ta;
  for (int i = 0; i < 12; i++) Ta.m[i] = T[i];
  MatchK mk{};
  SolveK sk0{};
  MH_TRY(upload_params(ctx, mk, sk0));
  hipLaunchKernelGGL(k_match_pl<false>, dim3(nblk(scan->n)), dim3(kBlock), 0, ctx->stream, ctx->d_state, Ta,
                     (float)distance_threshold, &ctx->d_params->mk, scan->x, scan->y, scan->z, (uint32_t)scan->n, map->view(),
                     ctx->pl_c.as<float4>(), ctx->pl_n.as<float4>(), (double*)nullptr, 0u);
  MH_HIP(hipGetLastError());
  mh_pairs_pl_out none{};
  uint64_t np = 0;
  MH_TRY(compact_pl_pairs(ctx, scan->n, out ? out : &none, mem, &np));
  if (info) info->n_pairs = np;
  return MH_OK;
}

mh_status mh_nn_search_pt2pl_knn(const mh_map* map, const mh_scan* scan, const double T[12], const mh_pt2pl_knn_params* params,
                                 const mh_pairs_pl_out* out, int32_t mem, mh_match_info* info) {
  MH_REQUIRE(map && scan && T && params, "null argument");
  MH_REQUIRE(mem == MH_MEM_HOST || mem == MH_MEM_DEVICE, "bad mem space");
  MH_REQUIRE(map->ctx->device == scan->ctx->device, "map and scan live on different devices");
  MH_REQUIRE(pose_ok(T), "non-finite pose");
  MH_REQUIRE(params->knn >= 3 && params->knn <= (uint32_t)kMaxPlaneKnn, "knn must be 3..MH_MAX_PLANE_KNN");
  MH_REQUIRE(isfinite(params->distance_threshold) && isfinite(params->plane_eigen_threshold) && isfinite(params->search_radius) &&
             params->search_radius > 0.0, "bad thresholds");
  mh_ctx* ctx = scan->ctx;
  MH_TRY(set_device(ctx));
  MH_TRY(map_ready_on(map, ctx->stream));
  if (info) {
    info->n_pairs = 0;
    info->potential_pairings = scan->n;
  }
  if (scan->n == 0) return MH_OK;
  MH_TRY(ensure_pl_buffers(ctx, scan->n));
  PoseArg Ta;
  for (int i = 0; i < 12; i++) Ta.m[i] = T[i];
  PlKnnArg a;
  a.distance_threshold = params->distance_threshold;
  a.plane_eigen_threshold = params->plane_eigen_threshold;
  a.radius2 = (float)(params->search_radius * params->search_radius);
  a.knn = params->knn;
  a.min_pts = params->minimum_plane_points < 3u ? 3u : params->minimum_plane_points;  // (three points span a plane)
  hipLaunchKernelGGL(k_match_pl_knn, dim3(nblk(scan->n)), dim3(kBlock), 0, ctx->stream, Ta, a, scan->x, scan->y, scan->z,
                     (uint32_t)scan->n, map->view(), ctx->pl_c.as<float4>(), ctx->pl_n.as<float4>());
  MH_HIP(hipGetLastError());
  mh_pairs_pl_out none{};
  uint64_t np = 0;
  MH_TRY(compact_pl_pairs(ctx, scan->n, out ? out : &none, mem, &np));
  if (info) info->n_pairs = np;
  return MH_OK;
}

mh_status mh_icp_get_pt2pl_pairs(const mh_scan* scan, const mh_pairs_pl_out* out, int32_t mem, uint64_t* n_pairs) {
  MH_REQUIRE(scan && out, "null argument");
  MH_REQUIRE(mem == MH_MEM_HOST || mem == MH_MEM_DEVICE, "bad mem space");
  mh_ctx* ctx = scan->ctx;
  MH_TRY(set_device(ctx));
  if (n_pairs) *n_pairs = 0;
  if (scan->n == 0 || !ctx->pl_c.p || ctx->pl_c.bytes < scan->n * sizeof(float4)) return MH_OK;  // no pt2pl matcher has run
  return compact_pl_pairs(ctx, scan->n, out, mem, n_pairs);
}

// ---- solver-granular entry points ---------------------------------------------------------------
namespace {
// stage 3 (or 6/9) SoA float arrays of n elements into one device buffer with a common stride
mh_status stage_soa(mh_ctx* ctx, DevBuf& buf, const float* const* arrs, int count, size_t n, int32_t mem, size_t* stride_out) {
  const size_t stride = ((n + 63) / 64) * 64;
  MH_TRY(buf.reserve((size_t)count * stride * sizeof(float) + 256));
  for (int a = 0; a < count; a++)
    MH_TRY(stage_in(ctx, buf, (size_t)a * stride * sizeof(float), arrs[a], n * sizeof(float), mem));
  *stride_out = stride;
  return MH_OK;
}
}  // namespace

mh_status mh_gn_solve(mh_ctx* ctx, const mh_pairs_pt2pt* pp, const mh_pairs_pt2pl* pl, int32_t mem,
                      const mh_gn_params* p, const mh_prior* prior, double T_io[12], int32_t* n_steps, int32_t* solver_ok,
                      mh_gn_step* trace) {
  MH_REQUIRE(ctx && p && T_io, "null argument");
  MH_REQUIRE(mem == MH_MEM_HOST || mem == MH_MEM_DEVICE, "bad mem space");
  MH_REQUIRE(p->max_inner_iterations >= 1 && p->max_inner_iterations <= kMaxGnTrace, "max_inner_iterations out of [1,16]");
  MH_REQUIRE(p->robust_kernel <= MH_KERNEL_GM_C2, "unknown robust kernel");
  MH_REQUIRE(pose_ok(T_io), "non-finite linearisation point");
  const size_t np = pp ? pp->n : 0, nl = pl ? pl->n : 0;
  MH_REQUIRE(np == 0 || (pp->lx && pp->ly && pp->lz && pp->gx && pp->gy && pp->gz), "null pt2pt arrays");
  MH_REQUIRE(nl == 0 || (pl->lx && pl->ly && pl->lz && pl->cx && pl->cy && pl->cz && pl->nx && pl->ny && pl->nz),
             "null pt2pl arrays");
  if (n_steps) *n_steps = 0;
  if (solver_ok) *solver_ok = 1;
  MH_TRY(set_device(ctx));
  MH_TRY(ensure_state(ctx));
  hipStream_t s = ctx->stream;
  MH_HIP(mh::wait_stream(s));
  // stage pairings: build_a = pt2pt (l xyz | g xyz), build_b = pt2pl (l | c | n)
  size_t sp = 0, sl = 0;
  if (np) {
    const float* arrs[6] = {pp->lx, pp->ly, pp->lz, pp->gx, pp->gy, pp->gz};
    MH_TRY(stage_soa(ctx, ctx->build_a, arrs, 6, np, mem, &sp));
  }
  if (nl) {
    const float* arrs[9] = {pl->lx, pl->ly, pl->lz, pl->cx, pl->cy, pl->cz, pl->nx, pl->ny, pl->nz};
    MH_TRY(stage_soa(ctx, ctx->build_b, arrs, 9, nl, mem, &sl));
  }
  MH_TRY(ensure_pair_buffers(ctx, np));
  const uint32_t nbp = np ? nblk(np) : 0, nbl = nl ? nblk(nl) : 0;
  MH_TRY(ctx->partials_b.reserve((size_t)kGenN * (nbl ? nbl : 1) * sizeof(double)));
  MH_TRY(ctx->trace.reserve(sizeof(mh_gn_step) * kMaxGnTrace));
  const float* L = ctx->build_a.as<float>();
  if (np)
    hipLaunchKernelGGL(k_pack_pairs, dim3(nbp), dim3(kBlock), 0, s, L + 3 * sp, (uint32_t)np, (uint32_t)sp,
                       ctx->pair_q.as<float4>(), ctx->pair_gidx.as<uint32_t>());
  init_state(ctx->h_state, T_io);
  ctx->h_state->cur_kparam = p->robust_kernel_param;  // solver-granular path: fixed robust-kernel parameter
  MH_HIP(hipMemcpyAsync(ctx->d_state, ctx->h_state, sizeof(IcpDeviceState), hipMemcpyHostToDevice, s));
  MatchK mk{};
  mk.kernel = p->robust_kernel;
  mk.w_pt2pt = p->weight_pt2pt;
  SolveK sk;
  memset(&sk, 0, sizeof(sk));
  sk.max_iterations = 1;
  sk.disable_stall = 1;
  sk.max_inner = p->max_inner_iterations;
  sk.min_delta = p->min_delta;
  sk.max_cost = p->max_cost;
  fill_prior(sk, prior);
  sk.gn_trace = (mh_gn_step*)ctx->trace.p;
  mk.use_fixed = 1;
  mk.kparam_fixed = p->robust_kernel_param;
  MH_TRY(upload_params(ctx, mk, sk));
  MH_HIP(hipMemsetAsync(ctx->trace.p, 0, sizeof(mh_gn_step) * kMaxGnTrace, s));
  const float* P = ctx->build_b.as<float>();
  for (uint32_t in = 0; in < p->max_inner_iterations; in++) {
    if (np)
      hipLaunchKernelGGL(k_accum<false>, dim3(nblk_acc(np)), dim3(kBlock), 0, s, ctx->d_state, in == 0 ? 1u : 0u, &ctx->d_params->mk,
                         L, L + sp, L + 2 * sp, (uint32_t)np, ctx->pair_q.as<float4>(),
                         ctx->pair_gidx.as<uint32_t>(), ctx->partials.as<double>(), nblk_acc(np));
    if (nl)
      hipLaunchKernelGGL(k_accum_pl, dim3(nbl), dim3(kBlock), 0, s, ctx->d_state, p->robust_kernel,
                         p->robust_kernel_param, p->weight_pt2pl, P, P + 3 * sl, P + 6 * sl, (uint32_t)nl, (uint32_t)sl,
                         ctx->partials_b.as<double>(), nbl);
    hipLaunchKernelGGL(k_solve, dim3(1), dim3(kSolveThreads), 0, s, ctx->d_state, &ctx->d_params->sk,
                       ctx->partials.as<double>(), np ? nblk_acc(np) : 0u, np ? nblk_acc(np) : 0u,
                       ctx->partials_b.as<double>(), nbl, nbl, in == 0 ? 1u : 0u);
  }
  MH_HIP(hipGetLastError());
  MH_HIP(hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(IcpDeviceState), hipMemcpyDeviceToHost, s));
  MH_HIP(mh::wait_stream(s));
  const IcpDeviceState* h = ctx->h_state;
  for (int i = 0; i < 12; i++) T_io[i] = h->T[i];
  if (n_steps) *n_steps = (int32_t)h->n_solves;
  if (solver_ok) *solver_ok = (int32_t)h->solver_ok;
  if (trace) MH_HIP(hipMemcpy(trace, ctx->trace.p, sizeof(mh_gn_step) * p->max_inner_iterations, hipMemcpyDeviceToHost));
  return MH_OK;
}

mh_status mh_covariance(mh_ctx* ctx, const mh_pairs_pt2pt* pp, const mh_pairs_pt2pl* pl, int32_t mem, const double T[12],
                        double findif_xyz, double findif_ang, double cov[36]) {
  MH_REQUIRE(ctx && T && cov, "null argument");
  MH_REQUIRE(mem == MH_MEM_HOST || mem == MH_MEM_DEVICE, "bad mem space");
  MH_REQUIRE(findif_xyz > 0 && findif_ang > 0, "finite-difference steps must be > 0");
  MH_REQUIRE(pose_ok(T), "non-finite pose");
  const size_t np = pp ? pp->n : 0, nl = pl ? pl->n : 0;
  for (int i = 0; i < 36; i++) cov[i] = (i % 7 == 0) ? 1e6 : 0.0;
  if (np + nl == 0) return MH_OK;  // "no pairings -> no estimation": diag(1e6)
  MH_TRY(set_device(ctx));
  MH_TRY(ensure_state(ctx));
  hipStream_t s = ctx->stream;
  MH_HIP(mh::wait_stream(s));
  size_t sp = 0, sl = 0;
  if (np) {
    const float* arrs[3] = {pp->lx, pp->ly, pp->lz};
    MH_TRY(stage_soa(ctx, ctx->build_a, arrs, 3, np, mem, &sp));
  }
  if (nl) {
    const float* arrs[6] = {pl->lx, pl->ly, pl->lz, pl->nx, pl->ny, pl->nz};
    MH_TRY(stage_soa(ctx, ctx->build_b, arrs, 6, nl, mem, &sl));
  }
  MH_TRY(ensure_pair_buffers(ctx, np));
  const uint32_t nbp = np ? nblk(np) : 0, nbl = nl ? nblk(nl) : 0;
  MH_TRY(ctx->partials_b.reserve((size_t)kGenN * (nbl ? nbl : 1) * sizeof(double)));
  init_state(ctx->h_state, T);
  MH_HIP(hipMemcpyAsync(ctx->d_state, ctx->h_state, sizeof(IcpDeviceState), hipMemcpyHostToDevice, s));
  if (np) MH_HIP(hipMemsetAsync(ctx->pair_gidx.p, 0, np * sizeof(uint32_t), s));  // all valid
  {
    MatchK mk0{};
    SolveK sk0{};
    sk0.cov_hx = findif_xyz;
    sk0.cov_ha = findif_ang;
    MH_TRY(upload_params(ctx, mk0, sk0));
  }
  hipLaunchKernelGGL(k_cov_prepare, dim3(1), dim3(64), 0, s, ctx->d_state, &ctx->d_params->sk, 1u);
  const float* L = ctx->build_a.as<float>();
  const float* P = ctx->build_b.as<float>();
  if (np)
    hipLaunchKernelGGL(k_cov_accum, dim3(nbp), dim3(kBlock), 0, s, ctx->d_state, 1u, L, L + sp, L + 2 * sp, (uint32_t)np,
                       ctx->pair_gidx.as<uint32_t>(), ctx->partials.as<double>(), nbp);
  if (nl)
    hipLaunchKernelGGL(k_cov_accum_pl, dim3(nbl), dim3(kBlock), 0, s, ctx->d_state, P, P + 3 * sl, (uint32_t)nl,
                       (uint32_t)sl, ctx->partials_b.as<double>(), nbl);
  hipLaunchKernelGGL(k_cov_finalize, dim3(1), dim3(kSolveThreads), 0, s, ctx->d_state, 1u, ctx->partials.as<double>(), nbp, nbp,
                     ctx->partials_b.as<double>(), nbl, nbl);
  MH_HIP(hipGetLastError());
  MH_HIP(hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(IcpDeviceState), hipMemcpyDeviceToHost, s));
  MH_HIP(mh::wait_stream(s));
  for (int i = 0; i < 36; i++) cov[i] = ctx->h_state->cov[i];
  return MH_OK;
}
