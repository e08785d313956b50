#!/usr/bin/env python3
"""The measured latency floor of the match step: how long does the ACCESS PATTERN of k_match4_b take
when the arithmetic is stripped?

    bash tools/build_floor.sh && python tools/match_floor.py --out profiles/r04_match_floor.json

Debug library (tools/libmolahip_floor.so = the product sources + -DMH_DEBUG_FLOOR):
  1. one lock-step alignment batch of the headline workload (32 x C2, a map and a scan per job, inputs resident) in which the
     launch of ICP iteration k writes down, per scan point, what its search touched: per probe batch the voxel code every lane
     of the quad probed, and the winner's record;
  2. the same batch again without the capture, every match launch timed with HIP events, and after it k_match_floor_b
     replayed 20 times back to back on the same job descriptors: same grid, same occupancy (8 waves per SIMD), the same
     DEPENDENT chain per quad -- point + previous pairing -> slot probes of a batch -> W records per lane and round trip of
     the merged ranges -> next batch -> winner's record -> pairing written -- and one compare per record instead of the fp64
     transform, the voxel bounds, the distances and the 64-bit keys.
floor / real = how close the kernel is to what its memory-access schedule alone costs on this hardware.
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import capi  # noqa: E402

REAL_ONLY = "--real-only" in sys.argv  # (child process: the PRODUCT library's own launch times under the same conditions)
if not REAL_ONLY:
    capi.LIB_PATH = os.path.join(ROOT, "tools", "libmolahip_floor.so")
import bench  # noqa: E402  (generate_inputs: the headline workload's 32 draws)

# (a "no transform" switch exists in the kernel but is no what-if: un-transformed points probe other, mostly empty voxels)
WHAT_IF = {"as_is": 0, "no_winner_fetch": 1, "no_previous_pairing_read": 2, "no_pairing_write": 8, "records_read_as_12_bytes": 16,
           "half_the_records": 32, "narrow_io_dword_per_lane": 64, "narrow_io_no_previous_pairing_read": 64 | 2,
           "narrow_io_no_pairing_write": 64 | 8, "narrow_io_no_winner_fetch": 64 | 1,
           "whole_voxels_no_sub_voxel_index": 128}  # the schedule before the sub-voxel index (profiles/r04_match_kernel.md, section 5)


def setup(S, ws):
    map_ctx = capi.Context(0)
    maps = [capi.Map(map_ctx, w.voxel_size, w.cap).build(w.map_xyz) for w in ws]
    ctxs = [capi.Context(0) for _ in ws]
    scans = [capi.Scan(c, w.scan_xyz) for c, w in zip(ctxs, ws)]
    guesses = [w.T_guess for w in ws]
    w0 = ws[0]
    params = capi.ICPParams(max_iterations=w0.n_iters, disable_stall_test=True, threshold=w0.threshold, kernel_param=w0.kernel_param,
                            poll_every=w0.n_iters, profile=2)
    return map_ctx, maps, ctxs, scans, guesses, params


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=32)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--iterations", default="1,5,10,19")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--real-only", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_match_floor.json"))
    args = ap.parse_args()
    S = args.jobs
    ws, _ = bench.generate_inputs(args.workload, list(range(S)))
    keep = setup(S, ws)
    _, maps, _, scans, guesses, params = keep
    w0 = ws[0]
    if args.real_only:
        for _ in range(2):
            capi.icp_align_batch(maps, scans, guesses, params)
        vals = []
        for _ in range(5):
            r = capi.icp_align_batch(maps, scans, guesses, params)
            vals.append(r[0]["match_kernel_ms"] * S / r[0]["n_match_launches"])
        print(json.dumps({"real_ms_avg_over_launches_product_library": float(np.median(vals))}))
        return
    import subprocess
    real = json.loads(subprocess.run([sys.executable, os.path.abspath(__file__), "--real-only", "--jobs", str(S), "--workload", args.workload],
                                     capture_output=True, text=True, timeout=900).stdout.strip().splitlines()[-1])
    real_ms = real["real_ms_avg_over_launches_product_library"]
    L = capi.lib()
    L.mh_debug_floor_setup.argtypes = [C.c_uint32] * 5
    L.mh_debug_floor_setup.restype = C.c_int
    L.mh_debug_floor_result.argtypes = [C.POINTER(C.c_double), C.c_void_p, C.c_size_t]
    L.mh_debug_floor_result.restype = C.c_int
    stride = max(len(w.scan_xyz) for w in ws)
    out = {"workload": "%d x %s in lock step (a map and a scan per job, inputs resident), %d ICP iterations" % (S, w0.name, w0.n_iters),
           "kernel": "k_match4_b of the PRODUCT library (real) vs k_match_floor_b of the debug library (the same dependent chain of loads and "
                     "the same address generation -- fp64 transform, voxel index, hash, merged ranges -- with one compare per record instead "
                     "of voxel bounds, distances, 64-bit keys and the threshold test)",
           "real_ms_avg_over_launches_product_library": real_ms,
           "replays_per_measurement": args.reps, "per_iteration": {}, "what_if": {}}
    for _ in range(2):  # warm-up (code objects, graphs, buffers)
        capi.icp_align_batch(maps, scans, guesses, params)
    res = (C.c_double * 8)()
    sizes = np.array([len(w.scan_xyz) for w in ws])
    valid = np.arange(stride)[None, :] < sizes[:, None]

    def measure(k, flags):
        best = None
        for _ in range(3):
            assert L.mh_debug_floor_setup(S, stride, 0x80000000 | k, args.reps, flags) == 0  # (bit 31: never an iteration -> no capture)
            capi.icp_align_batch(maps, scans, guesses, params)
            assert L.mh_debug_floor_result(res, None, 0) == 0
            best = res[0] if best is None else min(best, res[0])
        return best

    last_k = None
    for k in [int(v) for v in args.iterations.split(",")]:
        assert L.mh_debug_floor_setup(S, stride, k, 0, 0) == 0
        capi.icp_align_batch(maps, scans, guesses, params)          # capture run (its timings are perturbed: ignored)
        scripts = np.zeros((S, stride, 2), np.uint32)
        assert L.mh_debug_floor_result(res, scripts.ctypes.data_as(C.c_void_p), scripts.size) == 0
        nb = scripts[:, :, 1] >> 24
        got = (nb < 15) & valid
        first = scripts[:, :, 1] & 0xFFFFFF
        probes0 = sum((((first >> (6 * s)) & 63) != 63) for s in range(4))
        cur = {"floor_ms": measure(k, 0), "points_captured_share": float(got.sum() / valid.sum()),
               "probe_batches_per_point_mean": float(nb[got].mean()), "probe_batches_per_point_max": int(nb[got].max()),
               "slot_probes_in_first_batch_per_point_mean": float(probes0[got].mean()),
               "points_without_a_winner_share": float(((scripts[:, :, 0] == 0xFFFFFFFF) & got).sum() / max(1, got.sum()))}
        cur["floor_over_real"] = cur["floor_ms"] / real_ms
        out["per_iteration"][str(k)] = cur
        last_k = k
        print("[floor] iteration %2d: floor %.4f ms vs real %.4f ms (product library, launch average) -> %.3f; %.2f batches, %.2f probes in the "
              "first batch per point" % (k, cur["floor_ms"], real_ms, cur["floor_over_real"], cur["probe_batches_per_point_mean"],
                                         cur["slot_probes_in_first_batch_per_point_mean"]), file=sys.stderr, flush=True)
    # what costs what: the replay with one ingredient taken out at a time (scripts of the last captured iteration)
    for name, flags in WHAT_IF.items():
        ms = measure(last_k, flags)
        out["what_if"][name] = {"floor_ms": ms, "vs_as_is": ms / out["per_iteration"][str(last_k)]["floor_ms"]}
        print("[floor] what-if %-28s %.4f ms (%.3f of the replay as it is)" % (name, ms, out["what_if"][name]["vs_as_is"]), file=sys.stderr, flush=True)
    its = out["per_iteration"].values()
    out["floor_ms_mean"] = float(np.mean([v["floor_ms"] for v in its]))
    out["frac_of_floor"] = out["floor_ms_mean"] / real_ms
    out["note"] = ("frac_of_floor = mean floor / the product kernel's launch average under the same conditions: the share of k_match4_b's "
                   "time that its memory-access schedule and address generation alone account for.  The floor is optimistic by "
                   "construction (launched back to back: warmer caches than between the accumulate / solve launches of a real iteration; "
                   "iteration 0's un-bounded search is not replayed) and pessimistic in two respects: it reads 8 bytes of script per point, "
                   "and it narrows every batch's record ranges (the sub-voxel index) under the bound the previous pairing gives, where the "
                   "real search has a tighter bound from its second batch on.")
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ("floor_ms_mean", "real_ms_avg_over_launches_product_library", "frac_of_floor")}))


if __name__ == "__main__":
    main()
