"""Accuracy of the device odometry against GROUND TRUTH (not against the oracle): the one signal that is independent of
the restatement both sides of every parity test share.  The drive is the synthetic city of
mola_lidar_odometry_amd/synth_city.py -- cross streets, houses with gaps, parked cars, poles, trees; skewed 64 x 1875-ray
sweeps with per-point time stamps; the vehicle pulls away from rest and turns -- run through molahip-lo-cli (C++) with the
reference's default and NDT pipelines.  tools/accuracy_ablation.py is the long (1000-scan) version with every App. B
switch; profiles/r04_accuracy.json holds its numbers."""
import json
import os
import subprocess

import numpy as np
import pytest

from mola_lidar_odometry_amd import synth_city, trajectory

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "mola_lidar_odometry_amd", "molahip-lo-cli")
N_SCANS = 260  # ~150 m: the start from rest, a straight, the first corner


@pytest.fixture(scope="module")
def city_drive(tmp_path_factory):
    base = tmp_path_factory.mktemp("city")
    seq_dir, drive = synth_city.write_kitti_drive(str(base), N_SCANS, time_channel=True)
    return str(base), seq_dir, drive, synth_city.ground_truth_44(drive)


def _run(seq_dir, pipeline, out, env=None, time_field=True):
    cmd = [CLI, "--pipeline", os.path.join(ROOT, "pipelines", pipeline), "--seq-dir", seq_dir, "--out", out]
    if time_field:
        cmd += ["--time-field", "12"]
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    rep = next(json.loads(l) for l in r.stdout.splitlines() if l.startswith("{") and "sequence_dir" in l)
    _, est = trajectory.read_tum(out)
    return rep, est


@pytest.mark.gpu
def test_city_drive_ate_below_a_fifth_of_a_percent_of_the_path(city_drive):
    base, seq_dir, drive, gt = city_drive
    path = synth_city.path_length(drive["poses"])
    assert path > 120.0
    rep, est = _run(seq_dir, "lidar3d-default-hip.yaml", os.path.join(base, "default.tum"))
    assert rep["scans"] == N_SCANS and rep["good"] == N_SCANS - 1, rep  # every alignment accepted (the first scan has none)
    assert len(est) == N_SCANS
    ate = trajectory.ate_rmse(est, gt, "origin")
    # VERDICT r3 set ATE <= 0.5 % of the path, r4 asked for 0.2 % now that 0.05-0.1 % is what is measured
    assert ate <= 0.002 * path, (ate, path)
    assert trajectory.ate_rmse(est, gt, "se3") <= ate + 1e-9
    # a KITTI-like layer: a few thousand points into align(), a local map that keeps growing
    assert 800 <= rep["mean_icp_points"] <= 8000, rep
    assert rep["final_map_points"] > 150_000, rep
    # the de-skew filter matters on this drive (9 m/s: 0.9 m of motion inside a sweep): without it the same run is clearly worse
    rep2, est2 = _run(seq_dir, "lidar3d-default-hip.yaml", os.path.join(base, "noskew.tum"), env={"MOLA_SKIP_DESKEW": "true"})
    ate2 = trajectory.ate_rmse(est2, gt[:len(est2)], "origin")
    assert ate2 > 1.5 * ate, (ate, ate2)


@pytest.mark.gpu
def test_city_drive_ndt_pipeline_ate(city_drive):
    base, seq_dir, drive, gt = city_drive
    path = synth_city.path_length(drive["poses"])
    rep, est = _run(seq_dir, "lidar3d-ndt-hip.yaml", os.path.join(base, "ndt.tum"))
    assert rep["scans"] == N_SCANS and rep["good"] >= N_SCANS - 2, rep
    ate = trajectory.ate_rmse(est, gt[:len(est)], "origin")
    assert ate <= 0.002 * path, (ate, path)


@pytest.mark.gpu
def test_sixteen_sequences_in_one_process_reproduce_the_solo_trajectory(city_drive):
    """A sequence's trajectory must not depend on what it shared the device with: sixteen copies of the drive through one
    molahip-lo-cli process -- alignments merged into lock-step batches of varying composition, lone ones run singly, layers on
    both sides of the 2 k-point boundary between the batch chains -- give sixteen times the solo run's file, byte for byte.
    (Round 4: this is the check that caught data handed from one launch to the next being read stale at the next launch's
    very start -- k_step16's comment -- at a rate of a few alignments per run.)"""
    base, seq_dir, drive, gt = city_drive
    solo_out = os.path.join(base, "solo16.tum")
    _run(seq_dir, "lidar3d-default-hip.yaml", solo_out)
    solo = open(solo_out).read()
    cmd = [CLI, "--pipeline", os.path.join(ROOT, "pipelines", "lidar3d-default-hip.yaml"), "--time-field", "12", "--out", os.path.join(base, "many.tum")]
    for _ in range(16):
        cmd += ["--seq-dir", seq_dir]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-800:]
    reps = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{") and "sequence_dir" in l]
    assert len(reps) == 16
    differing = [k for k, q in enumerate(reps) if open(q["tum"]).read() != solo]
    assert not differing, differing


def test_city_generator_is_deterministic_and_on_the_road():
    """CPU: the plan is a pure function of its arguments, the route keeps clear of every building, sweeps have the size of
    a KITTI scan and all returns lie within the range limit."""
    a, b = synth_city.route_plan(300), synth_city.route_plan(300)
    assert np.array_equal(a["poses"], b["poses"]) and np.array_equal(a["twists"], b["twists"])
    city = synth_city.make_city()
    p = a["poses"].reshape(-1, 3, 4)[:, :, 3]
    bx = city["boxes"]
    ground = bx[:, 2] < 0.5  # (foliage clumps hang above the road)
    inside = ((p[:, None, 0] > bx[None, ground, 0] - 0.5) & (p[:, None, 0] < bx[None, ground, 3] + 0.5) &
              (p[:, None, 1] > bx[None, ground, 1] - 0.5) & (p[:, None, 1] < bx[None, ground, 4] + 0.5)).any(1)
    assert not inside.any()
    rc = synth_city.Raycaster(city)
    xyz, t = rc.sweep(a["poses"][200], a["twists"][200], seed=a["seeds"][200])
    xyz2, _ = rc.sweep(a["poses"][200], a["twists"][200], seed=a["seeds"][200])
    assert np.array_equal(xyz, xyz2)
    assert 100_000 < len(xyz) <= 120_000
    assert np.linalg.norm(xyz, axis=1).max() < 80.2 and abs(t).max() <= 0.05 + 1e-6
    # suspension: the attitude moves, and stays small
    R = a["poses"].reshape(-1, 3, 4)[:, :, :3]
    pitch = -np.arcsin(R[:, 2, 0])
    assert 1e-3 < np.abs(pitch[50:]).max() < 0.02
