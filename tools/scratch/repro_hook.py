import faulthandler, sys, os
faulthandler.enable()
sys.path.insert(0, os.getcwd())
import numpy as np
from mola_lidar_odometry_amd import _mp2p_icp_hip as hl, synth
w = synth.workload_small()
OUR_YAML = "pipelines/lidar3d-default-hip.yaml"
cfg = hl.Config.FromYamlFile(OUR_YAML)["icp_settings_with_vel"]
def _maps(hl, w):
    g = hl.metric_map_t()
    hv = hl.HashedVoxelPointCloud(w.voxel_size, w.cap)
    hv.setPoints(w.map_xyz)
    g.set_layer("localmap", hv)
    l = hl.metric_map_t()
    l.set_layer("decimated_for_icp", hl.PointCloud(w.scan_xyz))
    return l, g, hv
l, g, _ = _maps(hl, w)
guess = hl.TPose3D(*w.guess_ypr)
for mode in ("device", "host", "replay"):
    print("mode", mode, flush=True)
    icp, params = hl.icp_pipeline_from_yaml(cfg)
    src = hl.ParameterSource()
    src.updateVariable("ADAPTIVE_THRESHOLD_SIGMA", w.sigma)
    icp.attachToParameterSource(src)
    params.maxIterations = 40
    chk = hl.CPose3D(guess)
    if mode == "device":
        icp.setDeviceHook(0.15, float(np.deg2rad(0.75)), chk)
    else:
        def hook(it, T, chk=chk):
            print("  hook", it, flush=True)
            d = hl.CPose3D.from_matrix(T) - chk
            t = np.asarray(d.matrix()).reshape(3, 4)
            ang = np.arccos(np.clip((np.trace(t[:, :3]) - 1) / 2, -1, 1))
            return bool(np.linalg.norm(t[:, 3]) > 0.15 or ang > np.deg2rad(0.75))
        icp.setIterationHook(hook)
        icp.setHookReplay(mode == "replay")
    r = icp.align(l, g, guess, params)
    print("  ->", r.terminationReason.name, r.nIterations, flush=True)
