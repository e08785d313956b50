"""How much do the UNVERIFIED upstream choices move the result?  (the reference's arithmetic lives in
packages that are absent here, so a few behaviours are switches: the GemanMcClure weight form U1, the motion-model prior.)
Runs the CPU oracle driver over the synthetic drive once per variant and reports, per variant, the largest per-scan pose
difference from the default and the ATE against the drive's ground truth.  A variant whose spread is below the parity
tolerance (1e-4 m / 1e-4 rad) cannot be told apart by an A/B against the reference; one above it can -- that is the list
tools/parity_pin.py sweeps.  CPU only; writes profiles/<tag>_u_sensitivity.json."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import synth, trajectory  # noqa: E402
from oracle import odometry_oracle as oo  # noqa: E402

PIPE = os.path.join(ROOT, "pipelines", "lidar3d-default-hip.yaml")
VARIANTS = [("default (GemanMcClure c^4/(c^2+e^2)^2, no prior)", {}),
            ("U1 GemanMcClure_KISS  c^2/(c+e^2)^2", {"MOLA_HIP_ROBUST_KERNEL": "GemanMcClure_KISS"}),
            ("U1 GemanMcClure_Barron 1/(e^2/(4c^2)+1)^2", {"MOLA_HIP_ROBUST_KERNEL": "GemanMcClure_Barron"}),
            ("U1 GemanMcClure_C2    c^2/(c^2+e^2)^2", {"MOLA_HIP_ROBUST_KERNEL": "GemanMcClure_C2"}),
            ("Cauchy                c^2/(c^2+e^2)", {"MOLA_HIP_ROBUST_KERNEL": "Cauchy"}),
            ("motion-model prior on", {"MOLA_HIP_MOTION_MODEL_PRIOR": "true"})]


def run(drive, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        o = oo.OdometryOracle(PIPE, n_threads=8)
        for (xyz, t), st in zip(drive["scans"], drive["stamps"]):
            o.on_lidar(st, xyz, t)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return np.stack([r["pose"] for r in o.records]), [r["icp_iterations"] for r in o.records]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    drive = synth.make_drive(n)
    G = np.stack([trajectory.to44(p) for p in drive["poses"]])
    gt = np.linalg.inv(G[0])[None] @ G
    base, rows = None, []
    for name, env in VARIANTS:
        poses, iters = run(drive, env)
        P = np.stack([trajectory.to44(p) for p in poses])
        if base is None:
            base = P
        dt = np.linalg.norm(P[:, :3, 3] - base[:, :3, 3], axis=1)
        Rrel = np.einsum("nij,nkj->nik", P[:, :3, :3], base[:, :3, :3])
        dr = np.arccos(np.clip((np.trace(Rrel, axis1=1, axis2=2) - 1) / 2, -1, 1))
        rows.append({"variant": name, "env": env, "max_translation_diff_m": float(dt.max()), "max_rotation_diff_rad": float(dr.max()),
                     "ate_rmse_m": float(trajectory.ate_rmse(P, gt, "none")), "icp_iterations_total": int(sum(iters)),
                     "distinguishable_at_1e-4": bool(dt.max() > 1e-4 or dr.max() > 1e-4)})
        print("%-46s max dt %.3e m  max dr %.3e rad  ATE %.3f m  iterations %d" % (
            name, dt.max(), dr.max(), rows[-1]["ate_rmse_m"], rows[-1]["icp_iterations_total"]))
    out = os.path.join(ROOT, "profiles", tag + "_u_sensitivity.json")
    json.dump({"drive": "synth.make_drive(%d): street canyon, 32x600 sweeps, pulls away from rest" % n, "pipeline": "lidar3d-default-hip.yaml",
               "rows": rows}, open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
