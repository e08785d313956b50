"""development aid: an opaque iteration hook through the mirror of the mp2p_icp plugin API three ways -- the generic loop
(one host round trip per iteration, the hook called live), the replay on the fused device loop (molahip_host/hook_replay.h:
what the mp2p_icp adapter does) and, for pose-threshold hooks, the device-side hook -- with random hooks (stop at iteration
k, pose thresholds, never) and random guesses: same iteration count and termination; poses of the two fused variants
bitwise equal, the generic loop within 1e-9."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mola_lidar_odometry_amd import capi, synth  # noqa: E402

capi.lib()
from mola_lidar_odometry_amd import _mp2p_icp_hip as hl  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 19)
w = synth.workload_small()
cfg = hl.Config.FromYamlFile(os.path.join(ROOT, "pipelines", "lidar3d-default-hip.yaml"))["icp_settings_with_vel"]
g = hl.metric_map_t()
hv = hl.HashedVoxelPointCloud(w.voxel_size, w.cap)
hv.setPoints(w.map_xyz)
g.set_layer("localmap", hv)
bad = 0
for case in range(n_cases):
    n = int(rng.choice([300, 1200, len(w.scan_xyz)]))
    l = hl.metric_map_t()
    l.set_layer("decimated_for_icp", hl.PointCloud(w.scan_xyz[rng.permutation(len(w.scan_xyz))[:n]]))
    gy = w.guess_ypr + np.concatenate([rng.normal(0, 0.1, 3), rng.normal(0, 0.01, 3)])
    guess = hl.TPose3D(*gy)
    kind = str(rng.choice(["iteration", "pose", "never"]))
    stop_at = int(rng.integers(0, 12))
    tt, ta = float(rng.choice([0.05, 0.15, 0.4])), float(np.deg2rad(rng.choice([0.3, 0.75, 2.0])))
    max_it = int(rng.choice([6, 20, 40]))
    res = {}
    for mode in ("host", "replay") + (("device",) if kind == "pose" else ()):
        icp, params = hl.icp_pipeline_from_yaml(cfg)
        src = hl.ParameterSource()
        src.updateVariable("ADAPTIVE_THRESHOLD_SIGMA", w.sigma)
        icp.attachToParameterSource(src)
        params.maxIterations = max_it
        chk = hl.CPose3D(guess)
        if mode == "device":
            icp.setDeviceHook(tt, ta, chk)
        else:
            def hook(it, T, chk=chk):
                if kind == "iteration":
                    return it >= stop_at
                if kind == "never":
                    return False
                d = hl.CPose3D.from_matrix(T) - chk
                t = np.asarray(d.matrix()).reshape(3, 4)
                ang = np.arccos(np.clip((np.trace(t[:, :3]) - 1) / 2, -1, 1))
                return bool(np.linalg.norm(t[:, 3]) > tt or ang > ta)
            icp.setIterationHook(hook)
            icp.setHookReplay(mode == "replay")
        res[mode] = icp.align(l, g, guess, params)
    a, b = res["host"], res["replay"]
    ok = a.terminationReason.name == b.terminationReason.name and a.nIterations == b.nIterations and a.n_pairs() == b.n_pairs()
    ok = ok and float(np.abs(np.asarray(a.pose()) - np.asarray(b.pose())).max()) < 1e-9
    if "device" in res:
        c = res["device"]
        ok = ok and c.terminationReason.name == b.terminationReason.name and c.nIterations == b.nIterations
        ok = ok and np.array_equal(np.asarray(b.pose()), np.asarray(c.pose()))
    bad += 0 if ok else 1
    print("case %3d n=%5d hook=%-9s stop_at=%2d budget=%2d -> %s after %d iterations %s" % (
        case, n, kind, stop_at, max_it, b.terminationReason.name, b.nIterations, "ok" if ok else
        "MISMATCH host %s/%d replay %s/%d" % (a.terminationReason.name, a.nIterations, b.terminationReason.name, b.nIterations)), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
