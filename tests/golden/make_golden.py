"""Generates tests/golden/small_align.json: the per-iteration trajectory of ICP::align on the committed small
workload (synth.workload_small(): 2 000-point scan vs 20 000-point map, sigma=2 schedule, 20 fixed iterations, then
the yaml-default stall-terminated run), computed by the TWO independent CPU restatements -- the numpy oracle
(oracle/icp_oracle_np.py, float64, different formulations) for the first iterations and the C oracle for the full runs.
The reference itself cannot be run here (parity unpinned), so these vectors pin the restatement, not the
reference.  Run from the repo root:  python tests/golden/make_golden.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import synth  # noqa: E402
from oracle import icp_oracle_np as onp  # noqa: E402
from oracle import oracle_c  # noqa: E402

w = synth.workload_small()
om = oracle_c.Map(w.voxel_size, w.cap).insert(w.map_xyz)
fixed = oracle_c.icp_align(om, w.scan_xyz, w.T_guess, oracle_c.ICPParams(
    max_iterations=w.n_iters, disable_stall_test=True, threshold=w.threshold, kernel_param=w.kernel_param))
thr, kp = synth.threshold_schedule(w.sigma, 300)
stall = oracle_c.icp_align(om, w.scan_xyz, w.T_guess, oracle_c.ICPParams(max_iterations=300, threshold=thr, kernel_param=kp))

# independent numpy restatement on a 250-point subsample, 4 iterations (pure python: slow)
sub = w.scan_xyz[::8]
mp = onp.VoxelMap(w.voxel_size, w.cap).insert(w.map_xyz)
npy = onp.icp_align(mp, sub, w.T_guess, w.threshold, w.kernel_param, 4, disable_stall=True)

out = {
    "workload": w.name, "n_scan": int(len(w.scan_xyz)), "n_map": int(len(w.map_xyz)),
    "scan_checksum": float(np.float64(w.scan_xyz.astype(np.float64).sum())),
    "map_checksum": float(np.float64(w.map_xyz.astype(np.float64).sum())),
    "T_guess": [float(v) for v in w.T_guess],
    "fixed20": {"T_per_iteration": [[float(v) for v in t["T"]] for t in fixed["trace"]],
                "n_pairs_per_iteration": [t["n_pairs"] for t in fixed["trace"]],
                "T_final": [float(v) for v in fixed["T"]], "quality": fixed["quality"],
                "cov_diag": [float(fixed["cov"][i, i]) for i in range(6)]},
    "stall300": {"n_iterations": stall["n_iterations"], "termination": oracle_c.TERM_NAMES[stall["termination_reason"]],
                 "T_final": [float(v) for v in stall["T"]], "n_final_pairs": stall["n_final_pairs"]},
    "numpy_subsample_every8_4iters": {"T_per_iteration": [[float(v) for v in onp.T12(t["T"])] for t in npy["trace"]],
                                      "n_pairs_per_iteration": [t["n_pairs"] for t in npy["trace"]]},
}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "small_align.json"), "w"), indent=1)
print("written", out["stall300"], out["fixed20"]["n_pairs_per_iteration"][:3])
