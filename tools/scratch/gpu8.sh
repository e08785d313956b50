set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
# the six U12 failures of job 7, with their messages
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_host_layer.py -m gpu -q -k "skipping_plane_paired or ndt_pipeline_align_matches_oracle" 2>&1 | grep -E "^E |Error|assert|passed|failed|^tests" | head -60
# the step chain: parity suite with it switched on
MH_CHAIN_R=1 timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15
python - <<'PY'
import os, sys, json, subprocess, tempfile
sys.path.insert(0, os.getcwd())
import bench
from mola_lidar_odometry_amd import synth_city
tmp = tempfile.mkdtemp(prefix='molahip_st_')
seq, drive = synth_city.write_kitti_drive(tmp, 600, time_channel=True)
tums = {}
for name, env in (('old', {}), ('stepchain', {'MH_CHAIN_R': '1'}), ('stepchain_lead2', {'MH_CHAIN_R': '1', 'MH_STREAM_LEAD': '1'}), ('stepchain_lead4', {'MH_CHAIN_R': '1', 'MH_STREAM_LEAD': '3'}),
                  ('old_chunks', {'MH_NO_STREAM': '1'}), ('stepchain_chunks', {'MH_CHAIN_R': '1', 'MH_NO_STREAM': '1'})):
    for pipe in (bench.PIPELINE, bench.PIPELINE_NDT):
        for rep in range(2):
            per, prof, _ = bench.run_lo_cli(seq, 1, os.path.join(tmp, 'o'), pipeline=pipe, env=env)
            p = prof[0]
            print('CHAIN %-18s %-22s steady %.0f scans/s  onLidar %.4f ms  icp %.4f  enq %.1f exec %.1f polls %.2f' % (name, os.path.basename(pipe), per[0]['steady_scans_per_s'], p['onLidar'], p['onLidar.3.run_icp'], p['icp.enqueued_iterations'], p['icp.executed_iterations'], p['icp.host_polls']), flush=True)
        tums[(name, pipe)] = open(per[0]['tum']).read()
for pipe in (bench.PIPELINE, bench.PIPELINE_NDT):
    print('trajectory old == stepchain', os.path.basename(pipe), tums[('old', pipe)] == tums[('stepchain', pipe)], 'chunks:', tums[('stepchain', pipe)] == tums[('stepchain_chunks', pipe)])
# several sequences in one process
for name, env in (('old', {}), ('stepchain', {'MH_CHAIN_R': '1'})):
    for nseq in (4, 8, 16):
        per, prof, summ = bench.run_lo_cli(seq, nseq, os.path.join(tmp, 'm'), pipeline=bench.PIPELINE, env=env, max_scans=300)
        print('MULTI %-10s %2d sequences: %.0f scans/s' % (name, nseq, summ['steady_scans_per_s'] if summ else -1), flush=True)
PY
