// mh_k_match_rows.h -- Matcher_Point2Plane on NDT maps (lidar3d-ndt.yaml:195-200) and the row matcher (a DPP row of 16 lanes per
// point: k_match16 body, layers up to 32 k points), with the plane matcher riding along and the first accumulation fused in.
#pragma once

// ================================================================================================
// Matcher_Point2Plane on an NDT map (lidar3d-ndt.yaml:195-200): nearest planar voxel of the 27-block
// by centroid distance, accepted iff |n.(p'-c)| < threshold.  One lane per scan point; the 27 slot probes go out in
// three batches of nine unconditional loads, the nine centroid records of a batch likewise.
// ================================================================================================
__device__ __forceinline__ void acc_pt2pl_rows(double* v, const double* __restrict__ T, float lxf, float lyf, float lzf,
                                               const float4& c, const float4& nrm, uint32_t kernel, double kparam,
                                               double wpair) {
  const double lx = lxf, ly = lyf, lz = lzf;
  const double gx = T[0] * lx + T[1] * ly + T[2] * lz + T[3] - (double)c.x;
  const double gy = T[4] * lx + T[5] * ly + T[6] * lz + T[7] - (double)c.y;
  const double gz = T[8] * lx + T[9] * ly + T[10] * lz + T[11] - (double)c.z;
  const double nx = nrm.x, ny = nrm.y, nz = nrm.z;
  const double e = nx * gx + ny * gy + nz * gz;
  const double w = wpair * robust_weight(kernel, kparam, e * e);
  double J[6];
  J[0] = T[0] * nx + T[4] * ny + T[8] * nz;  // m = R^T n
  J[1] = T[1] * nx + T[5] * ny + T[9] * nz;
  J[2] = T[2] * nx + T[6] * ny + T[10] * nz;
  J[3] = ly * J[2] - lz * J[1];
  J[4] = lz * J[0] - lx * J[2];
  J[5] = lx * J[1] - ly * J[0];
  int q = 0;
#pragma unroll
  for (int a = 0; a < 6; a++)
#pragma unroll
    for (int b = a; b < 6; b++) v[q++] = w * J[a] * J[b];
#pragma unroll
  for (int a = 0; a < 6; a++) v[21 + a] = w * J[a] * e;
  v[27] = w * e * e;
  v[28] = 1.0;
}

// Matcher_Point2Plane's acceptance test.  thr > 0: point-to-plane distance |n.(p'-c)| < thr (default,
// MH_PT2PL_PLANE_DISTANCE); thr < 0 encodes MH_PT2PL_CENTROID_DISTANCE: |p'-c|^2 < thr^2, fp32, un-fused like the search.
__device__ __forceinline__ bool pl_accept(const f32x4& bn, float dx, float dy, float dz, float thr) {
  if (thr < 0.f) return (dx * dx + dy * dy) + dz * dz < thr * thr;
  return fabsf((bn.x * dx + bn.y * dy) + bn.z * dz) < thr;
}

template <bool FUSED>
__global__ __launch_bounds__(kBlock) void k_match_pl(const IcpDeviceState* __restrict__ st, PoseArg Targ, float thr_arg,
                                                     const MatchK* __restrict__ kp, const float* __restrict__ lx,
                                                     const float* __restrict__ ly, const float* __restrict__ lz, uint32_t n,
                                                     MapView map, float4* __restrict__ pl_c, float4* __restrict__ pl_n,
                                                     double* __restrict__ partials, uint32_t pstride) {
  __shared__ BlockSum<kGenN> lds;
  const MatchK k = *kp;
  double T[12];
  float thr;
  double kparam = 0.0;
  if (FUSED) {
    if (st->done) return;
    const uint32_t it = st->iter;
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = st->T[i];
    thr = (float)k.pl_thr[it];
    kparam = k.kparam[it];
  } else {
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = Targ.m[i];
    thr = thr_arg;
  }
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  double v[kGenN];
#pragma unroll
  for (int j = 0; j < kGenN; j++) v[j] = 0.0;
  if (i < n) {
    const float x = lx[i], y = ly[i], z = lz[i];
    float px, py, pz;
    transform_point(T, x, y, z, px, py, pz);
    const float lim = 1.0e6f;
    const bool valid = isfinite(px) && isfinite(py) && isfinite(pz) && fabsf(px * map.inv_vs) < lim &&
                       fabsf(py * map.inv_vs) < lim && fabsf(pz * map.inv_vs) < lim;
    float best = __builtin_inff();
    uint32_t best_first = 0;
    f32x4 bc = (f32x4)(0.f);
    if (valid) {
      const gslots_ptr slots4 = (gslots_ptr)map.slots;
      const gpts_ptr pts4 = (gpts_ptr)map.pts;
      const unsigned long long kbase = pack_key(voxel_of(px, map.inv_vs, map.trunc) - 1, voxel_of(py, map.inv_vs, map.trunc) - 1,
                                                voxel_of(pz, map.inv_vs, map.trunc) - 1);
#pragma unroll 1
      for (int ix = 0; ix < 3; ix++) {  // x outer: scan order is preserved for the first-minimum rule
        u32x4 sl[9];
        uint32_t first[9];
#pragma unroll
        for (int c = 0; c < 9; c++) {
          const unsigned long long key = kbase + ((unsigned long long)ix << 42) + ((unsigned long long)(c / 3) << 21) + (unsigned long long)(c % 3);
          sl[c] = slots4[hash_key(key) & map.mask];
        }
#pragma unroll
        for (int c = 0; c < 9; c++) {
          const unsigned long long key = kbase + ((unsigned long long)ix << 42) + ((unsigned long long)(c / 3) << 21) + (unsigned long long)(c % 3);
          unsigned long long sk = ((unsigned long long)sl[c].y << 32) | sl[c].x;
          if (sk != key && sk != kEmptyKey) {
            uint32_t h = hash_key(key) & map.mask;
            do {
              h = (h + 1) & map.mask;
              sl[c] = slots4[h];
              sk = ((unsigned long long)sl[c].y << 32) | sl[c].x;
            } while (sk != key && sk != kEmptyKey);
          }
          first[c] = (sk == key) ? sl[c].z : 0u;  // 0 = absent (a present voxel has first >= 2)
        }
        f32x4 cen[9];
#pragma unroll
        for (int c = 0; c < 9; c++) cen[c] = pts4[first[c] >= 2u ? first[c] - 2u : 0u];  // unconditional, clamped
#pragma unroll
        for (int c = 0; c < 9; c++)
          if (first[c] >= 2u && cen[c].w != 0.f) {
            const float dx = cen[c].x - px, dy = cen[c].y - py, dz = cen[c].z - pz;
            const float d2 = (dx * dx + dy * dy) + dz * dz;
            if (d2 < best) { best = d2; best_first = first[c]; bc = cen[c]; }
          }
      }
    }
    bool ok = false;
    f32x4 bn = (f32x4)(0.f);
    if (best_first >= 2u) {
      bn = ((gpts_ptr)map.pts)[best_first - 1u];
      const float dx = px - bc.x, dy = py - bc.y, dz = pz - bc.z;
      ok = pl_accept(bn, dx, dy, dz, thr);
    }
    const float4 c4 = make_float4(bc.x, bc.y, bc.z, ok ? 1.f : 0.f), n4 = make_float4(bn.x, bn.y, bn.z, 0.f);
    pl_c[i] = c4;
    pl_n[i] = n4;
    if (FUSED && ok) acc_pt2pl_rows(v, T, x, y, z, c4, n4, k.kernel, kparam, k.w_pt2pl);
  }
  if (FUSED) block_sum_rows<kGenN>(v, lds, partials, pstride, blockIdx.x);
}

// Matcher_Point2Plane with a DPP row (16 lanes) per point, for small layers.  k_match_pl walks the 27 voxels in three
// dependent groups of probes + centroid loads (57 us per launch on a 1 k-point layer); here lane r probes codes r and
// r + 16, reads the two statistics records of its voxels, and the row takes the minimum of (d2 to the centroid, code) --
// code order IS the reference's scan order -- in two round trips: nearest planar voxel of the 27-block by centroid
// distance (first in code order among equals), accepted iff |n.(p'-c)| < thr; every lane of the row returns the same
// centroid / normal / verdict.  Runs inside k_match16<true>.
__device__ __forceinline__ bool pl_row_search(const MapView& map, uint32_t r16, float px, float py, float pz, float thr,
                                              f32x4& bc, f32x4& bn) {
  const float lim = 1.0e6f;
  const bool valid = isfinite(px) && isfinite(py) && isfinite(pz) && fabsf(px * map.inv_vs) < lim &&
                     fabsf(py * map.inv_vs) < lim && fabsf(pz * map.inv_vs) < lim;
  nnkey_t best = kNNKeyNone;  // (d2 bits << 32 | code): first strict minimum in scan order
  f32x4 ca = (f32x4)(0.f), na = (f32x4)(0.f), cb = (f32x4)(0.f), nb = (f32x4)(0.f);
  if (valid) {  // row-uniform
    const gslots_ptr slots4 = (gslots_ptr)map.slots;
    const gpts_ptr pts4 = (gpts_ptr)map.pts;
    const unsigned long long kbase = pack_key(voxel_of(px, map.inv_vs, map.trunc) - 1, voxel_of(py, map.inv_vs, map.trunc) - 1,
                                              voxel_of(pz, map.inv_vs, map.trunc) - 1);
    const int code_a = (int)r16, code_b = (int)r16 + 16;
    const bool has_b = code_b < 27;
    const unsigned long long ka = nn_key_of(kbase, code_a), kb = nn_key_of(kbase, has_b ? code_b : code_a);
    const u32x4 sa = slots4[hash_key(ka) & map.mask];
    const u32x4 sb = slots4[hash_key(kb) & map.mask];
    uint32_t fa, cnt_a, fb, cnt_b;
    nn_resolve(map, slots4, ka, sa, true, fa, cnt_a);
    nn_resolve(map, slots4, kb, sb, has_b, fb, cnt_b);
    const bool pa = cnt_a > 0 || fa >= 2u, pb = has_b && (cnt_b > 0 || fb >= 2u);  // a present voxel has first >= 2
    // both statistics records of both voxels in one round trip (clamped, not predicated)
    ca = pts4[pa ? fa - 2u : 0u];
    na = pts4[pa ? fa - 1u : 0u];
    cb = pts4[pb ? fb - 2u : 0u];
    nb = pts4[pb ? fb - 1u : 0u];
    if (pa && ca.w != 0.f) {
      const float dx = ca.x - px, dy = ca.y - py, dz = ca.z - pz;
      const nnkey_t kk = ((nnkey_t)__float_as_uint((dx * dx + dy * dy) + dz * dz) << 32) | (uint32_t)code_a;
      best = kk < best ? kk : best;
    }
    if (pb && cb.w != 0.f) {
      const float dx = cb.x - px, dy = cb.y - py, dz = cb.z - pz;
      const nnkey_t kk = ((nnkey_t)__float_as_uint((dx * dx + dy * dy) + dz * dz) << 32) | (uint32_t)code_b;
      best = kk < best ? kk : best;
    }
  }
  best = row_min_key(best);
  const uint32_t wcode = nnkey_idx(best);
  bc = (f32x4)(0.f);
  bn = (f32x4)(0.f);
  bool ok = false;
  if (wcode != 0xFFFFFFFFu) {  // row-uniform: the owner lane hands its records to the row
    const bool from_b = wcode >= 16u;
    const uint32_t owner = wcode & 15u;
    const f32x4 mc = from_b ? cb : ca, mn = from_b ? nb : na;
    bc.x = __uint_as_float(row_bcast_u32(__float_as_uint(mc.x), owner));
    bc.y = __uint_as_float(row_bcast_u32(__float_as_uint(mc.y), owner));
    bc.z = __uint_as_float(row_bcast_u32(__float_as_uint(mc.z), owner));
    bn.x = __uint_as_float(row_bcast_u32(__float_as_uint(mn.x), owner));
    bn.y = __uint_as_float(row_bcast_u32(__float_as_uint(mn.y), owner));
    bn.z = __uint_as_float(row_bcast_u32(__float_as_uint(mn.z), owner));
    const float dx = px - bc.x, dy = py - bc.y, dz = pz - bc.z;
    ok = pl_accept(bn, dx, dy, dz, thr);
  }
  return ok;
}

// k_match16: a DPP row (16 lanes) per scan point for layers up to kRowMaxPoints (see nn_search_row16).
// PL: the same launch also runs Matcher_Point2Plane for the point (pl_row_search, pairings into pl_c / pl_n): the NDT
// pipeline's two matchers in one kernel instead of two.
// FUSED: the row leaders also accumulate the first Gauss-Newton step of their pairing and the workgroup writes one
// partial per row of sums (16 points per workgroup): layers of 2-32 k points -- what lidar3d-default.yaml really feeds --
// run match | solve | accumulate | solve, four launches per iteration instead of five.
template <bool PL, bool FUSED>
__device__ __forceinline__ void k_match16_body(const IcpDeviceState* __restrict__ st, const MatchK* __restrict__ kp,
                                                    const float* __restrict__ lx, const float* __restrict__ ly,
                                                    const float* __restrict__ lz, uint32_t n, MapView map,
                                                    float4* __restrict__ pair_q, uint32_t* __restrict__ pair_gidx,
                                                    float4* __restrict__ pl_c, float4* __restrict__ pl_n,
                                                    double* __restrict__ partials, uint32_t pstride) {
  __shared__ double rows[FUSED ? kAccN : 1][kBlock / 16 + 1];
  const uint32_t gl = blockIdx.x * kBlock + threadIdx.x;
  const uint32_t i = gl >> 4, r16 = gl & 15u;
  const uint32_t ic = i < n ? i : n - 1;
  const float x = G(lx)[ic], y = G(ly)[ic], z = G(lz)[ic];
  // state and parameters through the scalar path (uniform addresses, not written during this kernel: mh_nn_device.h)
  typedef const IcpDeviceState __attribute__((address_space(4))) * cstate_ptr;
  typedef const MatchK __attribute__((address_space(4))) * cmatchk_ptr;
  typedef const double __attribute__((address_space(4))) * cf64_ptr;
  const cstate_ptr cst = (cstate_ptr)uniform_const_ptr(st);
  const cmatchk_ptr ck = (cmatchk_ptr)uniform_const_ptr(kp);
  const uint32_t done = cst->done;
  // the record paired with this point under the previous pose bounds the search (k_match4_body has the story)
  const bool have_prev = cst->iter > 0 && !map.no_prev_bound;
  f32x4 prev = (f32x4){0.f, 0.f, 0.f, __builtin_inff()};
  if (have_prev) prev = G(reinterpret_cast<const f32x4*>(pair_q))[ic];  // grid-uniform branch
  double T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = cst->T[k];
  const float thr2 = cst->cur_thr2, ang2 = cst->cur_ang2;
  float pl_thr = 0.f;
  if (PL) pl_thr = (float)((cf64_ptr)uniform_const_ptr(ck->pl_thr))[cst->iter];
  uint32_t kernel = 0;
  double kparam = 0.0, wpair = 0.0;
  if (FUSED) {
    kernel = ck->kernel;
    wpair = ck->w_pt2pt;
    kparam = cst->cur_kparam;
  }
  if (done) return;              // grid-uniform
  if (!FUSED && i >= n) return;  // whole rows (FUSED: they stay for the barrier)
  Acc a;
  acc_zero(a);
  if (i < n) {  // row-uniform
    float px, py, pz;
    transform_point(T, x, y, z, px, py, pz);
    float bound0 = __builtin_inff();
    if (prev.w < __builtin_inff()) {
      const float dx = prev.x - px, dy = prev.y - py, dz = prev.z - pz;
      bound0 = (dx * dx + dy * dy) + dz * dz;  // the candidate arithmetic of the scans
    }
    const NNResult r = nn_search_row16(map, r16, px, py, pz, bound0);
    const float n2 = (px * px + py * py) + pz * pz;
    bool ok = r.found && (r.d2 < thr2 + ang2 * n2);
    if (PL) {  // Matcher_Point2Plane runs first in the reference's order; its verdict may keep the point out of the point matcher
      f32x4 bc, bn;
      const bool okp = pl_row_search(map, r16, px, py, pz, pl_thr, bc, bn);
      if (r16 == 0) {
        G(reinterpret_cast<f32x4*>(pl_c))[i] = (f32x4){bc.x, bc.y, bc.z, okp ? 1.f : 0.f};
        G(reinterpret_cast<f32x4*>(pl_n))[i] = (f32x4){bn.x, bn.y, bn.z, 0.f};
      }
      if (okp && ck->skip_pl_paired) ok = false;  // (the nearest point still goes to pair_q: it bounds the next search)
    }
    if (r16 == 0) {
      G(reinterpret_cast<f32x4*>(pair_q))[i] = (f32x4){r.pt.x, r.pt.y, r.pt.z, r.d2};
      G(pair_gidx)[i] = ok ? __float_as_uint(r.pt.w) : kNoMatch;
    }
    if (FUSED && r16 == 0) acc_pt2pt_masked(a, T, ok, x, y, z, r.pt.x, r.pt.y, r.pt.z, kernel, kparam, wpair);
  }
  if (FUSED) {  // 16 row leaders per workgroup -> one partial per sum, fixed order
    if (r16 == 0) {
#pragma unroll
      for (int j = 0; j < kAccN; j++) rows[j][threadIdx.x >> 4] = a.v[j];
    }
    __syncthreads();
    if (threadIdx.x < kAccN) {
      double sum = rows[threadIdx.x][0];
#pragma unroll
      for (int q = 1; q < (int)(kBlock / 16); q++) sum += rows[threadIdx.x][q];
      G(partials)[threadIdx.x * pstride + blockIdx.x] = sum;
    }
  }
}
