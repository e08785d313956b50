// hook_replay.h -- an arbitrary host-side iteration hook on top of the fused device loop.
//
// mp2p_icp::ICP::align [U] calls the user's iteration hook at the end of every iteration it completes (after the
// stall test) and stops with IterTermReason::HookRequest when the hook asks for it -- that is how mola::LidarOdometry
// gets its twist re-estimation loop (LidarOdometry.cpp:923-952 installs the hook, :958-1007 reacts to HookRequest).
// The fused loop of mh_icp_align never leaves the device, so a hook that is opaque code cannot run inside it.  But the
// iteration sequence does not depend on the hook (it only observes), so it can be REPLAYED: run the loop once with a
// per-iteration trace, feed the traced poses to the hook in order, and if it requests a stop at iteration k re-run the
// loop with a budget of k + 1 iterations -- deterministic reductions make the second run reproduce the first bit for bit
// up to k, and its final state (pose, pairings, quality, covariance) is what the hook-stopped loop would have left.
// Cost: one extra alignment per hook stop.  (The in-tree hook is a pure function of the pose and is evaluated on the
// device instead -- mh_icp_params::hook_* -- when the caller can say so; an adapter that only sees a std::function cannot.)
//
// Header-only and written against the C ABI alone so that the mp2p_icp adapter (host/adapters/mp2p_icp_plugin.cpp, needs
// MRPT to compile) and this repository's mirror classes (host/src/icp.cpp, compiled and tested here) share the logic.
#pragma once
#include <vector>

#include "molahip.h"

namespace molahip_host {

// run(max_iterations, trace /*may be null*/) -> mh_icp_result of an alignment WITHOUT device hook;
// hook(iteration, T /*12 doubles, pose after that iteration*/) -> true to request a stop.
template <class Run, class Hook>
mh_icp_result align_with_replayed_hook(uint32_t max_iterations, Run&& run, Hook&& hook) {
  std::vector<mh_icp_iter> trace(max_iterations ? max_iterations : 1);
  mh_icp_result r = run(max_iterations, trace.data());
  // iterations the reference's loop runs to their end (where the hook is called): all but a last one that broke out
  // with NoPairings / SolverError / Stalled, whose index is n_iterations
  const uint32_t n_hooked = r.n_iterations < max_iterations ? r.n_iterations : max_iterations;
  for (uint32_t k = 0; k < n_hooked; k++) {
    if (!hook(k, trace[k].T)) continue;
    if (!(k + 1 == n_hooked && r.termination_reason == MH_TERM_MAX_ITERATIONS)) r = run(k + 1, nullptr);
    r.termination_reason = MH_TERM_HOOK_REQUEST;
    r.n_iterations = k;  // the index of the iteration that broke out, as for every other early exit
    break;
  }
  return r;
}

}  // namespace molahip_host
