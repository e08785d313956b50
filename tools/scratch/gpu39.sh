set -x
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
python - <<'PY'
import os, sys, json, subprocess, tempfile
sys.path.insert(0, os.getcwd())
import bench
from mola_lidar_odometry_amd import synth_city
tmp = tempfile.mkdtemp(prefix='molahip_st_')
seq, drive = synth_city.write_kitti_drive(tmp, 400, time_channel=True)
tums = {}
for name, env in (('stepchain', {}), ('old', {'MH_NO_STEP_CHAIN': '1'})):
    for pipe in (bench.PIPELINE, bench.PIPELINE_NDT):
        for rep in range(2):
            per, prof, _ = bench.run_lo_cli(seq, 1, os.path.join(tmp, 'o'), pipeline=pipe, env=env)
            p = prof[0]
            print('CHAIN %-18s %-22s steady %.0f scans/s  onLidar %.4f ms  icp %.4f  enq %.1f exec %.1f polls %.2f' % (name, os.path.basename(pipe), per[0]['steady_scans_per_s'], p['onLidar'], p['onLidar.3.run_icp'], p['icp.enqueued_iterations'], p['icp.executed_iterations'], p['icp.host_polls']), flush=True)
            tums[(name, pipe)] = open(per[0]['tum']).read()
for pipe in (bench.PIPELINE, bench.PIPELINE_NDT):
    print('trajectory old == stepchain', os.path.basename(pipe), tums[('old', pipe)] == tums[('stepchain', pipe)])
for rep in range(5):
  for pipe in (bench.PIPELINE, bench.PIPELINE_NDT):
    for nseq in (8, 16, 16):
        try:
            per, prof, summ = bench.run_lo_cli(seq, nseq, os.path.join(tmp, 'm'), pipeline=pipe, max_scans=400)
        except Exception as e:
            print('MULTI FAILED', repr(e)[:300]); continue
        txt = [open(q['tum']).read().splitlines() for q in per]
        solo = tums[('stepchain', pipe)].splitlines()
        bad = [k for k, u in enumerate(txt) if u != solo[:len(u)]]
        print('MULTI %-22s %2d sequences: %.0f scans/s; sequences that differ from the solo run: %s' % (os.path.basename(pipe), nseq, summ['steady_scans_per_s'], bad), flush=True)
PY
