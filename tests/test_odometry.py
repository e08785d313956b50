"""Stand-alone odometry driver.

CPU: the pipeline files are recognised by the C++ driver (no GPU needed to load them), the trajectory I/O / metrics
are right, and the CPU oracle driver tracks a synthetic drive.  GPU (-m gpu): the C++ driver on libmolahip reproduces
the oracle driver scan by scan -- same decisions (key-frames, hook re-runs, iteration counts, layer sizes, map
sizes) and the same poses far inside the 1e-4 m / 1e-4 rad bar."""
import os

import numpy as np
import pytest

from mola_lidar_odometry_amd import synth, trajectory

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIPE = os.path.join(ROOT, "pipelines", "lidar3d-default-hip.yaml")
PIPE_NDT = os.path.join(ROOT, "pipelines", "lidar3d-ndt-hip.yaml")
REF_PIPES = "/root/reference/pipelines"


@pytest.fixture(scope="module")
def host():
    from mola_lidar_odometry_amd import _mp2p_icp_hip as H
    return H


@pytest.fixture(scope="module")
def drive():
    return synth.make_drive(14)


def _gt_rel(drive):
    G = np.stack([trajectory.to44(p) for p in drive["poses"]])
    return np.linalg.inv(G[0])[None] @ G


# ------------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize("path", [PIPE, PIPE_NDT, REF_PIPES + "/lidar3d-default.yaml", REF_PIPES + "/lidar3d-ndt.yaml"])
def test_driver_recognises_pipeline(host, path):
    if not os.path.exists(path):
        pytest.skip("reference tree not present on this box")
    lo = host.LidarOdometry()
    lo.initialize(host.Config.FromYamlFile(path))
    d = lo.describePipeline()
    assert d["layer_for_icp"] == "decimated_for_icp" and d["layer_for_map"] == "decimated_for_map"
    assert d["map_layer"] == "localmap" and d["min_points_to_filter"] == "2000"
    assert int(d["bbox_mode"]) == 1 and int(d["timestamp_method"]) == 1 and d["skip_deskew"] == "false"
    assert d["map_class"] == ("mola::NDT" if "ndt" in os.path.basename(path) else "mola::HashedVoxelPointCloud")
    assert d["formula:range_max"] == "1.2*ESTIMATED_SENSOR_MAX_RANGE"
    assert "ESTIMATED_SENSOR_MAX_RANGE" in d["formula:min_translation_between_keyframes"]
    with pytest.raises(RuntimeError):
        lo.initialize(host.Config.FromYamlFile(path))  # one object, one pipeline


def test_repo_pipelines_carry_the_reference_values(host):
    """The driver sections of pipelines/*-hip.yaml must say what the reference pipelines say (checked with PyYAML,
    independently of the C++ reader)."""
    if not os.path.exists(REF_PIPES):
        pytest.skip("reference tree not present on this box")
    from oracle import odometry_oracle as oo
    for mine, ref in ((PIPE, "lidar3d-default.yaml"), (PIPE_NDT, "lidar3d-ndt.yaml")):
        a, b = oo.load_pipeline(mine), oo.load_pipeline(os.path.join(REF_PIPES, ref))
        for key in ("observations_filter_adjust_timestamps", "observations_filter_1st_pass", "insert_observation_into_local_map"):
            assert a[key] == b[key], key
        deskew = [e for e in b["observations_filter_2nd_pass"] if e["class_name"].endswith("FilterDeskew")]
        assert a["observations_filter_2nd_pass"] == deskew
        for k, v in a["params"].items():
            if isinstance(v, dict):
                for kk, vv in v.items():
                    assert b["params"][k][kk] == vv, (k, kk)
            else:
                assert b["params"][k] == v, k
        assert a["navstate_fuse_params"]["max_time_to_use_velocity_model"] == b["navstate_fuse_params"]["max_time_to_use_velocity_model"]
        ma = a["localmap_generator"][0]["params"]["metric_map_definition"]
        mb = b["localmap_generator"][0]["params"]["metric_map_definition"]
        assert ma["class"] == mb["class"] and ma["creationOpts"] == mb["creationOpts"] and ma["insertOpts"] == mb["insertOpts"]


def test_unsupported_chain_is_rejected(host, tmp_path):
    txt = open(PIPE).read().replace("DecimateMethod::FirstPoint", "DecimateMethod::VoxelAverage")
    p = tmp_path / "bad.yaml"
    p.write_text(txt)
    with pytest.raises(RuntimeError, match="unsupported observation filter chain"):
        host.LidarOdometry().initialize(host.Config.FromYamlFile(str(p)))
    # ... while the alternative the reference's own pipeline file offers (lidar3d-default.yaml:292) is taken
    ok = tmp_path / "cta.yaml"
    ok.write_text(open(PIPE).read().replace("DecimateMethod::FirstPoint", "DecimateMethod::ClosestToAverage"))
    host.LidarOdometry().initialize(host.Config.FromYamlFile(str(ok)))


def test_tum_roundtrip_and_metrics(tmp_path):
    rng = np.random.default_rng(0)
    n = 400
    poses = np.tile(np.eye(4), (n, 1, 1))
    for i in range(1, n):
        d = trajectory.to44(synth.pose_from_ypr([2.5, 0.01 * rng.normal(), 0.0, 0.004 * rng.normal() + 0.002, 0.0, 0.0]))
        poses[i] = poses[i - 1] @ d
    stamps = 10.0 + 0.1 * np.arange(n)
    f = tmp_path / "t.tum"
    trajectory.write_tum(str(f), stamps, poses)
    s2, p2 = trajectory.read_tum(str(f))
    np.testing.assert_allclose(s2, stamps, atol=1e-9)
    np.testing.assert_allclose(p2, poses, atol=2e-8)
    assert trajectory.ate_rmse(p2, poses) < 1e-7
    te, re, k = trajectory.kitti_relative_errors(poses, poses)
    assert k > 0 and te < 1e-9 and re < 1e-6
    # a 1 % scale error along the path shows up as ~1 % translation error
    est = poses.copy()
    est[:, :3, 3] *= 1.01
    te, re, _ = trajectory.kitti_relative_errors(est, poses)
    assert 0.9 < te < 1.1 and re < 1e-6
    # rigidly displaced copy: zero after SE(3) alignment, not before
    M = trajectory.to44(synth.pose_from_ypr([5, -3, 1, 0.4, 0.1, -0.2]))
    moved = M[None] @ poses
    assert trajectory.ate_rmse(moved, poses, "se3") < 1e-6 < trajectory.ate_rmse(moved, poses, "none")
    ia, ib = trajectory.associate(stamps[::2] + 0.001, stamps)
    np.testing.assert_array_equal(ib, np.arange(0, n, 2))
    # the reference's own GT fragment parses (23 poses, starts at identity)
    ref = "/root/reference/test/rslidar_fragment_gt.tum"
    if os.path.exists(ref):
        s, p = trajectory.read_tum(ref)
        assert len(s) == len(p) > 10 and np.allclose(p[0], np.eye(4), atol=1e-6)


def test_oracle_driver_tracks_synthetic_drive(drive):
    from oracle import odometry_oracle as oo
    o = oo.OdometryOracle(PIPE, n_threads=8)
    for (xyz, t), st in zip(drive["scans"], drive["stamps"]):
        r = o.on_lidar(st, xyz, t)
    recs = o.records
    assert recs[0]["first_scan"] and not recs[0]["icp_run"] and recs[0]["map_updated"]
    assert recs[1]["icp_run"] and not recs[1]["had_motion_model"] and not recs[1]["map_updated"]  # App. C.6
    assert all(r["icp_good"] and r["had_motion_model"] for r in recs[2:])
    assert sum(r["twist_corrections"] for r in recs) >= 1  # the hook loop is exercised
    est = np.stack([trajectory.to44(p) for _, p in o.trajectory])
    assert trajectory.ate_rmse(est, _gt_rel(drive), "none") < 0.2
    sig = [r["sigma"] for r in recs[1:]]
    assert all(0.1 <= s <= 3.0 for s in sig) and sig[-1] < sig[0]  # adaptive threshold shrinks while tracking well
    # dropped scan: closer in time than min_time_between_scans
    r = o.on_lidar(drive["stamps"][-1] + 1e-4, *drive["scans"][-1])
    assert r["dropped"]


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_driver_matches_oracle_driver(host, drive, tmp_path):
    from oracle import odometry_oracle as oo
    o = oo.OdometryOracle(PIPE, n_threads=8)
    lo = host.LidarOdometry()
    lo.initialize(host.Config.FromYamlFile(PIPE))
    worst_t = worst_r = 0.0
    for k, ((xyz, t), st) in enumerate(zip(drive["scans"], drive["stamps"])):
        a = lo.onLidar(st, xyz, t)
        b = o.on_lidar(st, xyz, t)
        for key in ("dropped", "first_scan", "icp_run", "icp_good", "had_motion_model", "map_updated", "restarted",
                    "icp_iterations", "twist_corrections", "align_calls", "termination", "n_raw", "n_for_map",
                    "n_for_icp", "n_map_points", "n_map_voxels"):
            assert a[key] == b[key], (k, key, a[key], b[key])
        for key in ("goodness", "sigma", "estimated_sensor_max_range", "instantaneous_sensor_max_range",
                    "decim_map_resolution", "decim_icp_resolution"):
            assert abs(a[key] - b[key]) <= 1e-9 * max(1.0, abs(b[key])), (k, key, a[key], b[key])
        np.testing.assert_allclose(a["twist"], b["twist"], rtol=0, atol=1e-6)
        Ta, Tb = np.array(a["pose"]).reshape(3, 4), b["pose"].reshape(3, 4)
        worst_t = max(worst_t, float(np.linalg.norm(Ta[:, 3] - Tb[:, 3])))
        worst_r = max(worst_r, float(np.linalg.norm(Ta[:, :3] - Tb[:, :3])))
    assert worst_t < 1e-6 and worst_r < 1e-6, (worst_t, worst_r)  # bar: 1e-4 m / 1e-4 rad
    # trajectory out, TUM file, accuracy against the ground truth of the synthetic drive
    f = tmp_path / "est.tum"
    lo.saveTrajectoryTUM(str(f))
    stamps, est = trajectory.read_tum(str(f))
    np.testing.assert_allclose(stamps, drive["stamps"], atol=1e-6)
    assert trajectory.ate_rmse(est, _gt_rel(drive), "none") < 0.2
    dropped = lo.onLidar(drive["stamps"][-1] + 1e-4, *drive["scans"][-1])
    assert dropped["dropped"]
    # [n,4] records (x,y,z,intensity as in a KITTI .bin) take the same path as [n,3] points
    lo2 = host.LidarOdometry()
    lo2.initialize(host.Config.FromYamlFile(PIPE))
    for (xyz, t), st in list(zip(drive["scans"], drive["stamps"]))[:4]:
        lo2.onLidar(st, np.concatenate([xyz, np.ones((len(xyz), 1), np.float32)], 1), t)
    for ra, rb in zip(lo2.records(), lo.records()[:4]):
        assert ra["pose"] == rb["pose"] and ra["n_for_icp"] == rb["n_for_icp"]
    # ... and so do records that carry the fields elsewhere, time stamp included: [intensity, y, t, x, z]
    lo3 = host.LidarOdometry()
    lo3.initialize(host.Config.FromYamlFile(PIPE))
    for (xyz, t), st in list(zip(drive["scans"], drive["stamps"]))[:4]:
        rec = np.stack([np.ones(len(xyz), np.float32), xyz[:, 1], t, xyz[:, 0], xyz[:, 2]], 1)
        lo3.onLidar(st, rec, xyz_fields=(3, 1, 4), t_field=2)
    for ra, rb in zip(lo3.records(), lo.records()[:4]):
        assert ra["pose"] == rb["pose"] and ra["n_for_icp"] == rb["n_for_icp"]


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["both", "second"])
def test_hip_driver_matches_oracle_driver_with_closest_to_average_decimation(host, drive, tmp_path, which):
    """decimate_method: DecimateMethod::ClosestToAverage (the commented alternative of lidar3d-default.yaml:292) in the chain's
    decimations: the device driver and the oracle driver make the same layers, decisions and poses."""
    from oracle import odometry_oracle as oo
    txt = open(PIPE).read()
    assert txt.count("DecimateMethod::FirstPoint") == 2
    if which == "both":
        txt = txt.replace("DecimateMethod::FirstPoint", "DecimateMethod::ClosestToAverage")
    else:
        head, tail = txt.rsplit("DecimateMethod::FirstPoint", 1)
        txt = head + "DecimateMethod::ClosestToAverage" + tail
    pipe = tmp_path / "cta.yaml"
    pipe.write_text(txt)
    o = oo.OdometryOracle(str(pipe), n_threads=8)
    lo = host.LidarOdometry()
    lo.initialize(host.Config.FromYamlFile(str(pipe)))
    ref = host.LidarOdometry()
    ref.initialize(host.Config.FromYamlFile(PIPE))
    differs = False
    for k, ((xyz, t), st) in enumerate(list(zip(drive["scans"], drive["stamps"]))[:8]):
        a = lo.onLidar(st, xyz, t)
        b = o.on_lidar(st, xyz, t)
        c = ref.onLidar(st, xyz, t)
        for key in ("icp_run", "icp_good", "map_updated", "icp_iterations", "align_calls", "termination", "n_raw", "n_for_map",
                    "n_for_icp", "n_map_points", "n_map_voxels"):
            assert a[key] == b[key], (k, key, a[key], b[key])
        Ta, Tb = np.array(a["pose"]).reshape(3, 4), b["pose"].reshape(3, 4)
        assert np.linalg.norm(Ta - Tb) < 1e-6, (k, Ta, Tb)
        differs = differs or a["pose"] != c["pose"]
    assert differs  # (the method is not a no-op: other survivors, other poses than FirstPoint's)


def test_oracle_driver_motion_model_prior_and_initial_twist(drive, monkeypatch):
    """navstate_fuse_params of the -hip pipelines: `motion_model_prior` hands the motion model's covariance (last ICP
    covariance + (sigma_random_walk_acceleration * dt)^2, inverted) to align() as LidarOdometry.cpp:859-861 does;
    MOLA_INITIAL_VX (eval/cli_kitti.sh:25) gives a motion model from the second scan on."""
    from oracle import odometry_oracle as oo
    monkeypatch.setenv("MOLA_HIP_MOTION_MODEL_PRIOR", "true")
    monkeypatch.setenv("MOLA_INITIAL_VX", "1.5")
    o = oo.OdometryOracle(PIPE, n_threads=8)
    assert o.motion_model_prior and o.initial_twist is not None and o.initial_twist[0] == 1.5
    for (xyz, t), st in list(zip(drive["scans"], drive["stamps"]))[:6]:
        o.on_lidar(st, xyz, t)
    assert o.records[1]["had_motion_model"]          # from the initial twist
    est, info = o._nav_estimate(drive["stamps"][6])[0], o._nav_estimate(drive["stamps"][6])[2]
    assert info is not None and np.allclose(info, info.T) and np.all(np.linalg.eigvalsh(info) > 0)
    # dt = 0.1 s: position information ~ 1 / (0.1 m)^2, orientation ~ 1 / (1 rad)^2 (sigmas 1.0 m/s^2, 10 rad/s^2)
    assert 50 < info[0, 0] <= 100.0 + 1e-6 and 0.5 < info[5, 5] <= 1.0 + 1e-9


@pytest.mark.gpu
def test_hip_driver_matches_oracle_driver_with_motion_model_prior(host, drive, monkeypatch):
    """The same scan-by-scan comparison with the prior term switched on (and an initial twist): the driver now exercises
    the prior path of mh_icp_align on every scan; records identical to the oracle driver's, poses within 1e-6."""
    from oracle import odometry_oracle as oo
    monkeypatch.setenv("MOLA_HIP_MOTION_MODEL_PRIOR", "true")
    monkeypatch.setenv("MOLA_INITIAL_VX", "1.5")
    o = oo.OdometryOracle(PIPE, n_threads=8)
    lo = host.LidarOdometry()
    lo.initialize(host.Config.FromYamlFile(PIPE))
    worst = 0.0
    for k, ((xyz, t), st) in enumerate(zip(drive["scans"], drive["stamps"])):
        a = lo.onLidar(st, xyz, t)
        b = o.on_lidar(st, xyz, t)
        for key in ("icp_run", "icp_good", "had_motion_model", "map_updated", "icp_iterations", "twist_corrections",
                    "align_calls", "termination", "n_for_icp", "n_map_points"):
            assert a[key] == b[key], (k, key, a[key], b[key])
        worst = max(worst, float(np.abs(np.array(a["pose"]) - b["pose"]).max()))
    assert worst < 1e-6, worst
    assert lo.records()[1]["had_motion_model"]


@pytest.mark.gpu
def test_prefetch_overlap_gives_identical_records(host, drive):
    """Announcing scan k+1 before registering scan k (upload + first filter pass on a second stream, worker thread) must
    not change a single record; every announced scan but the first is picked up from the prepared layers."""
    def run(prefetch):
        lo = host.LidarOdometry()
        lo.initialize(host.Config.FromYamlFile(PIPE))
        items = [(st, np.ascontiguousarray(xyz, np.float32), np.ascontiguousarray(t, np.float32))
                 for (xyz, t), st in zip(drive["scans"], drive["stamps"])]
        for k, (st, xyz, t) in enumerate(items):
            if prefetch and k + 1 < len(items):
                lo.prefetch(items[k + 1][1], items[k + 1][2])
            lo.onLidar(st, xyz, t)
        return lo.records(), lo.profile()
    ra, pa = run(False)
    rb, pb = run(True)
    assert len(ra) == len(rb)
    for a, b in zip(ra, rb):
        assert a == b
    assert pb.get("prefetch_hits", 0) >= len(ra) - 2 and pb.get("prefetch_misses", 0) == 0
    assert "prefetch_hits" not in pa
    # a prefetch that is never picked up (different scan registered next) is harmless
    lo = host.LidarOdometry()
    lo.initialize(host.Config.FromYamlFile(PIPE))
    (x0, t0), (x1, t1) = drive["scans"][0], drive["scans"][1]
    lo.onLidar(drive["stamps"][0], x0, t0)
    lo.prefetch(np.ascontiguousarray(x0), np.ascontiguousarray(t0))
    r = lo.onLidar(drive["stamps"][1], x1, t1)
    assert r["pose"] == ra[1]["pose"]
    # a scan the filters reject (voxel index out of the key range) fails when it is registered, prefetched or not,
    # and the driver carries on with the next good scan
    bad = np.ascontiguousarray(drive["scans"][2][0]).copy()
    bad[5] = [3.0e9, 0.0, 0.0]
    tb = np.ascontiguousarray(drive["scans"][2][1])
    lo.prefetch(bad, tb)
    (x2, t2) = drive["scans"][2]
    lo.onLidar(drive["stamps"][2], x2, t2)  # launches the worker for `bad`
    with pytest.raises(RuntimeError):
        lo.onLidar(drive["stamps"][3], bad, tb)
    (x4, t4) = drive["scans"][4]
    assert lo.onLidar(drive["stamps"][4], x4, t4)["icp_run"]
    # the worker reads the caller's memory: arrays that would need a hidden converted copy are rejected, not converted
    with pytest.raises(RuntimeError):
        lo.prefetch(x4.astype(np.float64), t4)
    with pytest.raises(RuntimeError):
        lo.prefetch(np.asfortranarray(x4), t4)


@pytest.mark.gpu
def test_hip_driver_ndt_pipeline_and_restart(host, drive):
    """lidar3d-ndt: NDT local map (min-distance insertion, plane statistics), point-to-plane + point-to-point ICP --
    again scan by scan against the oracle driver."""
    from oracle import odometry_oracle as oo
    o = oo.OdometryOracle(PIPE_NDT, n_threads=8)
    lo = host.LidarOdometry()
    lo.initialize(host.Config.FromYamlFile(PIPE_NDT))
    for k, ((xyz, t), st) in enumerate(zip(drive["scans"][:8], drive["stamps"][:8])):
        a = lo.onLidar(st, xyz, t)
        b = o.on_lidar(st, xyz, t)
        for key in ("icp_run", "icp_good", "map_updated", "icp_iterations", "twist_corrections", "termination",
                    "n_for_map", "n_for_icp", "n_map_points", "n_map_voxels"):
            assert a[key] == b[key], (k, key, a[key], b[key])
        assert abs(a["goodness"] - b["goodness"]) < 1e-12 and abs(a["sigma"] - b["sigma"]) < 1e-9
        np.testing.assert_allclose(np.array(a["pose"]), b["pose"], rtol=0, atol=1e-6)
    recs = lo.records()
    assert all(r["icp_good"] for r in recs[1:]) and recs[-1]["n_map_points"] > 1000
    est = np.stack([trajectory.to44(p) for _, p in lo.trajectory()])
    assert trajectory.ate_rmse(est, _gt_rel(drive)[:8], "none") < 0.3
    # a hopeless second scan (unrelated cloud): ICP rejected right after the start => map dropped, start over (:1146-1156)
    lo2 = host.LidarOdometry()
    lo2.initialize(host.Config.FromYamlFile(PIPE))
    lo2.onLidar(1.0, *drive["scans"][0])
    rng = np.random.default_rng(5)
    junk = (rng.normal(0, 1, (20000, 3)) * [30, 30, 30] + [0, 0, 500]).astype(np.float32)
    r = lo2.onLidar(1.1, junk, None)
    assert r["icp_run"] and not r["icp_good"] and r["restarted"] and len(lo2.trajectory()) == 0
    r = lo2.onLidar(1.2, *drive["scans"][1])
    assert r["first_scan"] and r["map_updated"]


@pytest.mark.gpu
def test_sequence_runner_reports_metrics(tmp_path, capsys):
    import json
    from mola_lidar_odometry_amd import run_odometry
    run_odometry.main(["--synthetic", "8", "--rings", "32", "--azimuths", "600", "--out-dir", str(tmp_path)])
    lines = [json.loads(l) for l in capsys.readouterr().out.strip().splitlines()]
    seq, summary = lines[0], lines[-1]
    assert seq["scans"] == 8 and seq["good"] == 7 and seq["ate_rmse_m"] < 0.2 and os.path.exists(seq["tum"])
    assert summary["summary"] and summary["scans"] == 8 and summary["scans_per_s"] > 1.0


def _write_kitti_tree(root, drive, seq="00"):
    """The synthetic drive as a KITTI odometry tree: sequences/00/{velodyne/*.bin, times.txt, calib.txt}, poses/00.txt
    (camera-frame ground truth through the usual Tr, so the runner's frame conversion is exercised too)."""
    d = os.path.join(root, "sequences", seq)
    os.makedirs(os.path.join(d, "velodyne"))
    os.makedirs(os.path.join(root, "poses"))
    Tr = np.array([[0, -1, 0, 0.1], [0, 0, -1, -0.2], [1, 0, 0, 0.3], [0, 0, 0, 1.0]])  # velodyne -> camera
    for k, (xyz, _) in enumerate(drive["scans"]):
        np.concatenate([xyz, np.zeros((len(xyz), 1), np.float32)], 1).astype(np.float32).tofile(
            os.path.join(d, "velodyne", "%06d.bin" % k))
    np.savetxt(os.path.join(d, "times.txt"), drive["stamps"] - drive["stamps"][0], fmt="%.6e")
    with open(os.path.join(d, "calib.txt"), "w") as f:
        f.write("P0: " + " ".join(["0"] * 12) + "\nTr: " + " ".join("%.9e" % v for v in Tr[:3].reshape(-1)) + "\n")
    G = np.stack([trajectory.to44(p) for p in drive["poses"]])
    G = np.linalg.inv(G[0])[None] @ G
    cam = Tr[None] @ G @ np.linalg.inv(Tr)[None]
    np.savetxt(os.path.join(root, "poses", seq + ".txt"), cam[:, :3].reshape(len(cam), 12), fmt="%.9e")


def test_kitti_tree_reader_roundtrip(tmp_path, drive):
    from mola_lidar_odometry_amd import run_odometry
    _write_kitti_tree(str(tmp_path), drive)
    seq = os.path.join(str(tmp_path), "sequences", "00")
    scans = list(run_odometry._kitti_scans(seq))
    assert len(scans) == len(drive["scans"])
    np.testing.assert_array_equal(scans[3][1][:, :3], drive["scans"][3][0])  # rows are x,y,z,intensity
    assert abs(scans[3][0] - 0.3) < 1e-6 and scans[3][2] is None
    gt = run_odometry._kitti_gt(str(tmp_path), "00", run_odometry._kitti_calib_Tr(seq))
    np.testing.assert_allclose(gt, _gt_rel(drive), atol=1e-6)  # camera-frame poses come back in the velodyne frame
    t = trajectory.kitti_azimuth_timestamps(drive["scans"][0][0])
    assert t.dtype == np.float32 and abs(float(t.max()) - 0.05) < 2e-3 and abs(float(t.min()) + 0.05) < 2e-3


@pytest.mark.gpu
def test_sequence_runner_on_a_kitti_tree(tmp_path, drive, capsys):
    import json
    from mola_lidar_odometry_amd import run_odometry
    _write_kitti_tree(str(tmp_path / "kitti"), drive)
    run_odometry.main(["--kitti-root", str(tmp_path / "kitti"), "--seqs", "00", "--out-dir", str(tmp_path / "out")])
    seq = json.loads(capsys.readouterr().out.strip().splitlines()[0])
    # no per-point time stamps in a KITTI scan: the driver skips the de-skew (silently_ignore_no_timestamps), the
    # vehicle moves 0.8 m per sweep, so this is a plumbing check (frames, files, metrics), not an accuracy figure
    assert seq["sequence"] == "00" and seq["scans"] == len(drive["scans"]) and seq["good"] >= seq["scans"] - 2
    assert os.path.exists(seq["tum"]) and seq["ate_rmse_m"] < 1.5


@pytest.mark.gpu
def test_driver_objects_release_their_device_memory(host, drive):
    """Five drivers one after the other (each with its own maps, layers and scratch): device memory in use after the
    last one equals what it was after the first (handles are freed, scratch buffers do not accumulate)."""
    import gc
    from mola_lidar_odometry_amd import capi
    probe = capi.Context(0)

    def free_bytes():
        return probe.memory_info()[0]

    def one_run():
        lo = host.LidarOdometry()
        lo.initialize(host.Config.FromYamlFile(PIPE))
        for (xyz, t), st in zip(drive["scans"][:6], drive["stamps"][:6]):
            lo.onLidar(st, xyz, t)
        poses = [p for _, p in lo.trajectory()]
        del lo
        gc.collect()
        return poses, free_bytes()

    first_poses, free_after_first = one_run()
    for _ in range(4):
        poses, free_now = one_run()
        assert poses == first_poses  # and bitwise the same trajectory every time
    assert free_after_first - free_now < 8 << 20, (free_after_first, free_now)


def test_native_cli_is_built_and_explains_itself():
    """build() also produces molahip-lo-cli (the role of mola-lidar-odometry-cli in eval/cli_kitti.sh); without a GPU
    it can only print its usage."""
    import subprocess
    exe = os.path.join(ROOT, "mola_lidar_odometry_amd", "molahip-lo-cli")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "--seq-dir" in r.stderr


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_sequences_in_one_process_share_lockstep_batches(host, tmp_path):
    """Several sequences on ONE GPU from ONE process (what eval/cli_kitti.sh:23 does with processes): every LidarOdometry
    runs on its own host thread and its alignments join the others' in one mh_icp_align_batch per round
    (mp2p_icp_hip::AlignBatcher; per-job parameters, lock-step kernels).  Every sequence's records are those of its solo
    run -- different drives of different lengths, so the batch shrinks as sequences end -- and so are the TUM files the
    native command line writes for several --seq-dir."""
    import json
    import subprocess
    import threading
    drives = [synth.make_drive(n, seed=s, speed=v) for n, s, v in ((12, 4242, 8.0), (9, 777, 5.0), (14, 99, 10.0))]
    solo = []
    for d in drives:
        lo = host.LidarOdometry()
        lo.initialize(host.Config.FromYamlFile(PIPE))
        for (xyz, t), st in zip(d["scans"], d["stamps"]):
            lo.onLidar(st, xyz, t)
        solo.append(lo.records())
    batcher = host.AlignBatcher(len(drives))
    los, errors = [], []
    shared = host.LidarOdometry()
    shared.initialize(host.Config.FromYamlFile(PIPE))
    with pytest.raises(RuntimeError):  # threads must not share the process-wide default context
        shared.setAlignBatcher(batcher)
    for _ in drives:
        lo = host.LidarOdometry(own_context=True)
        lo.initialize(host.Config.FromYamlFile(PIPE))
        lo.setAlignBatcher(batcher)
        los.append(lo)

    def work(lo, d):
        try:
            for (xyz, t), st in zip(d["scans"], d["stamps"]):
                lo.onLidar(st, xyz, t)
        except Exception as e:  # noqa: BLE001
            errors.append(e)
        finally:
            batcher.leave()

    th = [threading.Thread(target=work, args=(lo, d)) for lo, d in zip(los, drives)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=60)
    assert not any(t.is_alive() for t in th), "a sequence thread is stuck"
    assert not errors, errors
    assert batcher.jobs() >= sum(len(d["scans"]) - 1 for d in drives) and batcher.batches() < batcher.jobs()
    for lo, ref in zip(los, solo):
        got = lo.records()
        assert len(got) == len(ref)
        for a, b in zip(got, ref):
            for key in ("pose", "icp_iterations", "twist_corrections", "align_calls", "termination", "goodness", "sigma",
                        "n_for_icp", "n_map_points", "map_updated", "icp_good"):
                assert a[key] == b[key], key
    # the same through the native command line: three --seq-dir in one process against three solo runs
    exe = os.path.join(ROOT, "mola_lidar_odometry_amd", "molahip-lo-cli")
    dirs = []
    for k, d in enumerate(drives):
        _write_kitti_tree(str(tmp_path / ("k%d" % k)), d)
        dirs.append(str(tmp_path / ("k%d" % k) / "sequences" / "00"))
    args = [exe, "--pipeline", PIPE, "--out", str(tmp_path / "multi.tum")]
    for d in dirs:
        args += ["--seq-dir", d]
    r = subprocess.run(args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    summary = json.loads(r.stdout.strip().splitlines()[-1])
    assert summary["sequences"] == 3 and summary["scans"] == sum(len(d["scans"]) for d in drives)
    # alignments AND the next scans' filter chains ran as batches over the sequences (mh_icp_align_batch,
    # mh_scan_preprocess_batch): fewer calls than jobs, every scan's filters accounted for
    assert summary["batches"] < summary["batches"] * summary["jobs_per_batch"]
    assert summary["filter_jobs"] >= summary["scans"] - 3 and summary["filter_batches"] < summary["filter_jobs"]
    for k, d in enumerate(dirs):
        one = str(tmp_path / ("solo%d.tum" % k))
        r1 = subprocess.run([exe, "--pipeline", PIPE, "--seq-dir", d, "--out", one], capture_output=True, text=True, timeout=300)
        assert r1.returncode == 0, r1.stderr
        assert open(one).read() == open(str(tmp_path / ("multi_%d.tum" % k))).read()


@pytest.mark.gpu
def test_native_cli_matches_the_python_runner(tmp_path, drive, capsys):
    """One KITTI-style sequence through the C++ command-line driver (next-scan prefetch on) and through the Python
    runner: the two TUM files are the same text."""
    import json
    import subprocess
    from mola_lidar_odometry_amd import run_odometry
    _write_kitti_tree(str(tmp_path / "kitti"), drive)
    seq_dir = str(tmp_path / "kitti" / "sequences" / "00")
    exe = os.path.join(ROOT, "mola_lidar_odometry_amd", "molahip-lo-cli")
    out_cli = str(tmp_path / "cli.tum")
    r = subprocess.run([exe, "--pipeline", PIPE, "--seq-dir", seq_dir, "--out", out_cli], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    line = next(json.loads(l) for l in r.stdout.splitlines() if l.startswith("{") and "sequence_dir" in l)  # (the summary line follows it)
    assert line["scans"] == len(drive["scans"]) and line["good"] >= line["scans"] - 2 and line["scans_per_s"] > 0
    run_odometry.main(["--kitti-root", str(tmp_path / "kitti"), "--seqs", "00", "--out-dir", str(tmp_path / "out")])
    seq = json.loads(capsys.readouterr().out.strip().splitlines()[0])
    assert open(out_cli).read() == open(seq["tum"]).read()
    seq_np = str(tmp_path / "np.tum")
    r2 = subprocess.run([exe, "--pipeline", PIPE, "--seq-dir", seq_dir, "--out", seq_np, "--no-prefetch"], capture_output=True,
                        text=True, timeout=300)
    assert r2.returncode == 0 and open(seq_np).read() == open(out_cli).read()
