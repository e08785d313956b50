// plugin_switches.h -- the unverified-upstream behaviours as process-wide environment switches.
//
// In plugin mode the pipeline file is the reference's own (pipelines/make_mola_hip.py changes one class name), and the
// upstream solver / matcher classes parse it with their own enums -- `robustKernel: 'RobustKernel::GemanMcClure'`
// (lidar3d-default.yaml:188) cannot name a variant of this library.  So the candidates are chosen through MOLA_HIP_*
// environment variables, read ONCE per process, and tools/parity_pin.py sweeps them against the reference's own run.
// Header-only and written against the C ABI alone: the mp2p_icp adapter (host/adapters/mp2p_icp_plugin.cpp, needs MRPT to
// compile) and this repository's mirror classes (host/src/icp.cpp, compiled and tested here) share it.
//
//   variable                  values (first = default)                                               SURVEY App. B
//   MOLA_HIP_ROBUST_KERNEL    GemanMcClure | GemanMcClure_KISS | GemanMcClure_Barron | GemanMcClure_C2 | Cauchy   U1
//   MOLA_HIP_INDEX_MODE       floor | trunc                                                                   U2/U3
//   MOLA_HIP_COV_STEP_XYZ     1e-7            MOLA_HIP_COV_STEP_ANG   1e-7                                    U7
//   MOLA_HIP_MIN_DELTA        1e-7            MOLA_HIP_MAX_COST       0                                       U8
//   MOLA_HIP_PT2PL_MODE       plane | centroid                                                                U10
//   MOLA_HIP_FAR_VOXEL_METRIC chebyshev | l1 | l2     (device-owned maps: remove_voxels_farther_than, yaml:238) a8
//   MOLA_HIP_FORCE_CPU        0 | 1                   (adapter only: every call to the upstream loop)
#pragma once
#include <cstdlib>
#include <cstring>
#include <string>

#include "molahip.h"

namespace molahip_host {

struct PluginSwitches {
  uint32_t gm_form = MH_KERNEL_GM_C4;  // what upstream's RobustKernel::GemanMcClure computes
  uint32_t index_mode = MH_INDEX_FLOOR;
  double cov_step_xyz = 1e-7, cov_step_ang = 1e-7;
  double min_delta = 1e-7, max_cost = 0.0;
  uint32_t pt2pl_mode = MH_PT2PL_PLANE_DISTANCE;
  uint32_t far_voxel_metric = MH_FAR_CHEBYSHEV;
  uint32_t matched_points = MH_MATCHED_POINTS_PAIR_AGAIN;  // MOLA_HIP_MATCHED_POINTS = again | skip (U12)
  bool force_cpu = false;
  // which of them came from the environment (the mirror classes only override their YAML values for those)
  bool has_gm_form = false, has_index_mode = false, has_cov_step = false, has_min_delta = false, has_max_cost = false,
       has_pt2pl_mode = false, has_far_metric = false;
};

/** The GPU this process's MRPT-side adapters run on: MOLA_HIP_DEVICE (default 0).  eval/cli_kitti.sh:23-36 runs one
 *  mola-lidar-odometry-cli process per sequence under GNU parallel; giving every job slot its own MOLA_HIP_DEVICE spreads
 *  them over a node's GPUs. */
inline int device_index() {
  const char* e = getenv("MOLA_HIP_DEVICE");
  return e ? atoi(e) : 0;
}

inline bool parse_gm_form(const char* s, uint32_t& out) {
  const char* p = strstr(s, "::");  // "RobustKernel::GemanMcClure_KISS" and "GemanMcClure_KISS" alike
  while (p) {
    s = p + 2;
    p = strstr(s, "::");
  }
  if (!strcmp(s, "GemanMcClure") || !strcmp(s, "GemanMcClure_C4")) out = MH_KERNEL_GM_C4;
  else if (!strcmp(s, "GemanMcClure_KISS")) out = MH_KERNEL_GM_KISS;
  else if (!strcmp(s, "GemanMcClure_Barron")) out = MH_KERNEL_GM_BARRON;
  else if (!strcmp(s, "GemanMcClure_C2")) out = MH_KERNEL_GM_C2;
  else if (!strcmp(s, "Cauchy")) out = MH_KERNEL_CAUCHY;
  else if (!strcmp(s, "None")) out = MH_KERNEL_NONE;
  else return false;
  return true;
}

/** Parse the environment (exposed separately from the cached accessor so that tests can call it after setenv). */
inline PluginSwitches read_plugin_switches() {
  PluginSwitches s;
  if (const char* e = getenv("MOLA_HIP_ROBUST_KERNEL")) s.has_gm_form = parse_gm_form(e, s.gm_form);
  if (const char* e = getenv("MOLA_HIP_INDEX_MODE")) {
    s.has_index_mode = true;
    s.index_mode = (!strcmp(e, "trunc") || !strcmp(e, "1")) ? MH_INDEX_TRUNC : MH_INDEX_FLOOR;
  }
  if (const char* e = getenv("MOLA_HIP_COV_STEP_XYZ")) { s.cov_step_xyz = atof(e); s.has_cov_step = true; }
  if (const char* e = getenv("MOLA_HIP_COV_STEP_ANG")) { s.cov_step_ang = atof(e); s.has_cov_step = true; }
  if (const char* e = getenv("MOLA_HIP_MIN_DELTA")) { s.min_delta = atof(e); s.has_min_delta = true; }
  if (const char* e = getenv("MOLA_HIP_MAX_COST")) { s.max_cost = atof(e); s.has_max_cost = true; }
  if (const char* e = getenv("MOLA_HIP_PT2PL_MODE")) {
    s.has_pt2pl_mode = true;
    s.pt2pl_mode = (!strcmp(e, "centroid") || !strcmp(e, "1")) ? MH_PT2PL_CENTROID_DISTANCE : MH_PT2PL_PLANE_DISTANCE;
  }
  if (const char* e = getenv("MOLA_HIP_FAR_VOXEL_METRIC")) {
    s.has_far_metric = true;
    s.far_voxel_metric = !strcmp(e, "l1") ? MH_FAR_L1 : !strcmp(e, "l2") ? MH_FAR_L2 : MH_FAR_CHEBYSHEV;
  }
  if (const char* e = getenv("MOLA_HIP_MATCHED_POINTS"))
    s.matched_points = (!strcmp(e, "skip") || !strcmp(e, "1")) ? MH_MATCHED_POINTS_SKIP : MH_MATCHED_POINTS_PAIR_AGAIN;
  if (const char* e = getenv("MOLA_HIP_FORCE_CPU")) s.force_cpu = atoi(e) != 0;
  return s;
}

inline PluginSwitches& plugin_switches_storage() {
  static PluginSwitches s = read_plugin_switches();
  return s;
}
inline const PluginSwitches& plugin_switches() { return plugin_switches_storage(); }
/** Re-read the environment (tests; a sweep driver that changes the variables inside one process). */
inline void reload_plugin_switches() { plugin_switches_storage() = read_plugin_switches(); }

/** MH_KERNEL_* for the NAME of an upstream mp2p_icp::RobustKernel enumerator [U] (names, not numeric values: the
 *  upstream enum's values are not relied on).  "GemanMcClure" resolves to the switched form. */
inline uint32_t kernel_from_upstream_name(const char* name, const PluginSwitches& sw) {
  uint32_t k = MH_KERNEL_NONE;
  if (!parse_gm_form(name, k)) return MH_KERNEL_NONE;
  if (k == MH_KERNEL_GM_C4) return sw.gm_form;
  return k;
}

/** MH_TERM_* -> an IterTermReason-like enum class E by enumerator NAME (upstream's mp2p_icp::IterTermReason [U] and the
 *  mirror's mp2p_icp_hip::IterTermReason both have these enumerators; their numeric values need not agree). */
template <class E>
E term_reason_to(uint32_t mh_term) {
  switch (mh_term) {
    case MH_TERM_NO_PAIRINGS: return E::NoPairings;
    case MH_TERM_SOLVER_ERROR: return E::SolverError;
    case MH_TERM_MAX_ITERATIONS: return E::MaxIterations;
    case MH_TERM_STALLED: return E::Stalled;
    case MH_TERM_QUALITY_CHECKPOINT_FAILED: return E::QualityCheckpointFailed;
    case MH_TERM_HOOK_REQUEST: return E::HookRequest;
    default: return E::Undefined;
  }
}

/** Apply the switches that were set in the environment to a parameter block built from the pipeline file. */
inline void apply_switches(mh_icp_params& ip, const PluginSwitches& sw) {
  if (sw.has_cov_step) { ip.cov_findif_xyz = sw.cov_step_xyz; ip.cov_findif_ang = sw.cov_step_ang; }
  if (sw.has_min_delta) ip.gn.min_delta = sw.min_delta;
  if (sw.has_max_cost) ip.gn.max_cost = sw.max_cost;
  if (sw.has_pt2pl_mode) ip.pt2pl_mode = sw.pt2pl_mode;
  ip.matched_points = sw.matched_points;
  if (sw.has_gm_form && ip.gn.robust_kernel == MH_KERNEL_GM_C4) ip.gn.robust_kernel = sw.gm_form;
}

}  // namespace molahip_host
