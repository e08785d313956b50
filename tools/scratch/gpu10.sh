set -x
cd $GRAFT_REPO_ROOT
MH_LOOP16=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "align" 2>&1 | tail -8
python - <<'PY'
import os, sys, json, subprocess, tempfile
sys.path.insert(0, os.getcwd())
import bench
from mola_lidar_odometry_amd import synth_city
tmp = tempfile.mkdtemp(prefix='molahip_st_')
seq, drive = synth_city.write_kitti_drive(tmp, 600, time_channel=True)
tums = {}
for name, env in (('old', {}), ('stepchain', {'MH_CHAIN_R': '1'}), ('loop16', {'MH_LOOP16': '1'}), ('loop16_32wg', {'MH_LOOP16': '1', 'MH_LOOP16_WGS': '32'}), ('loop16_16wg', {'MH_LOOP16': '1', 'MH_LOOP16_WGS': '16'})):
    for pipe in (bench.PIPELINE, bench.PIPELINE_NDT):
        for rep in range(2):
            try:
                per, prof, _ = bench.run_lo_cli(seq, 1, os.path.join(tmp, 'o'), pipeline=pipe, env=env)
            except Exception as e:
                print('CHAIN', name, 'FAILED', repr(e)[:300]); break
            p = prof[0]
            print('CHAIN %-18s %-22s steady %.0f scans/s  onLidar %.4f ms  icp %.4f  enq %.1f exec %.1f polls %.2f' % (name, os.path.basename(pipe), per[0]['steady_scans_per_s'], p['onLidar'], p['onLidar.3.run_icp'], p['icp.enqueued_iterations'], p['icp.executed_iterations'], p['icp.host_polls']), flush=True)
            tums[(name, pipe)] = open(per[0]['tum']).read()
for pipe in (bench.PIPELINE, bench.PIPELINE_NDT):
    for k in ('loop16', 'loop16_32wg', 'loop16_16wg', 'old'):
        if (k, pipe) in tums and ('stepchain', pipe) in tums:
            print('trajectory stepchain ==', k, os.path.basename(pipe), tums[('stepchain', pipe)] == tums[(k, pipe)])
PY
