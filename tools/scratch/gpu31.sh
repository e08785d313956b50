set -x
cd $GRAFT_REPO_ROOT
MH_ITER16=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_accuracy.py tests/test_host_layer.py -m gpu -q -x 2>&1 | tail -5
python - <<'PY'
import os, sys, json, subprocess, tempfile
sys.path.insert(0, os.getcwd())
import bench
from mola_lidar_odometry_amd import synth_city
tmp = tempfile.mkdtemp(prefix='molahip_st_')
seq, drive = synth_city.write_kitti_drive(tmp, 600, time_channel=True)
tums = {}
for name, env in (('step16', {}), ('iter16', {'MH_ITER16': '1'}), ('iter16 lead 2', {'MH_ITER16': '1', 'MH_STREAM_LEAD': '1'}), ('iter16 lead 4', {'MH_ITER16': '1', 'MH_STREAM_LEAD': '3'})):
    for pipe in (bench.PIPELINE, bench.PIPELINE_NDT):
        for rep in range(2):
            per, prof, _ = bench.run_lo_cli(seq, 1, os.path.join(tmp, 'o'), pipeline=pipe, env=env)
            p = prof[0]
            print('CHAIN %-18s %-22s steady %.0f scans/s  onLidar %.4f ms  icp %.4f  enq %.1f exec %.1f polls %.2f' % (name, os.path.basename(pipe), per[0]['steady_scans_per_s'], p['onLidar'], p['onLidar.3.run_icp'], p['icp.enqueued_iterations'], p['icp.executed_iterations'], p['icp.host_polls']), flush=True)
            tums[(name, pipe)] = open(per[0]['tum']).read()
for pipe in (bench.PIPELINE, bench.PIPELINE_NDT):
    a, b = tums[('step16', pipe)].splitlines(), tums[('iter16', pipe)].splitlines()
    import numpy as np
    A = np.array([[float(x) for x in l.split()] for l in a]); B = np.array([[float(x) for x in l.split()] for l in b])
    print('trajectory step16 vs iter16', os.path.basename(pipe), 'identical text:', a == b, 'max |d| %.3e' % np.abs(A - B).max())
for name, env in (('step16', {}), ('iter16', {'MH_ITER16': '1'})):
  for pipe in (bench.PIPELINE, bench.PIPELINE_NDT):
    for nseq in (4, 8, 16):
        per, prof, summ = bench.run_lo_cli(seq, nseq, os.path.join(tmp, 'm'), pipeline=pipe, max_scans=400, env=env)
        txt = [open(q['tum']).read().splitlines() for q in per]
        solo = tums[(name, pipe)].splitlines()
        bad = [k for k, u in enumerate(txt) if u != solo[:len(u)]]
        print('MULTI %-8s %-22s %2d sequences: %.0f scans/s; sequences that differ from the solo run: %s' % (name, os.path.basename(pipe), nseq, summ['steady_scans_per_s'], bad), flush=True)
PY
