set -x
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
python - <<'PY'
import os, sys, json, subprocess, tempfile
sys.path.insert(0, os.getcwd())
import bench
from mola_lidar_odometry_amd import synth_city
tmp = tempfile.mkdtemp(prefix='molahip_st_')
seq, drive = synth_city.write_kitti_drive(tmp, 600, time_channel=True)
tums = {}
for name, env in (('stepchain', {}), ('old', {'MH_NO_STEP_CHAIN': '1'}), ('stepchain', {})):
    for pipe in (bench.PIPELINE, bench.PIPELINE_NDT):
        for rep in range(2):
            per, prof, _ = bench.run_lo_cli(seq, 1, os.path.join(tmp, 'o'), pipeline=pipe, env=env)
            p = prof[0]
            print('CHAIN %-18s %-22s steady %.0f scans/s  onLidar %.4f ms  icp %.4f  enq %.1f exec %.1f polls %.2f' % (name, os.path.basename(pipe), per[0]['steady_scans_per_s'], p['onLidar'], p['onLidar.3.run_icp'], p['icp.enqueued_iterations'], p['icp.executed_iterations'], p['icp.host_polls']), flush=True)
            tums[(name, pipe)] = open(per[0]['tum']).read()
for pipe in (bench.PIPELINE, bench.PIPELINE_NDT):
    print('trajectory old == stepchain', os.path.basename(pipe), tums[('old', pipe)] == tums[('stepchain', pipe)])
for pipe in (bench.PIPELINE, bench.PIPELINE_NDT):
    for nseq in (8, 16):
        per, prof, summ = bench.run_lo_cli(seq, nseq, os.path.join(tmp, 'm'), pipeline=pipe, max_scans=400)
        txt = [open(q['tum']).read().splitlines() for q in per]
        solo = tums[('stepchain', pipe)].splitlines()
        bad = [k for k, u in enumerate(txt) if u != solo[:len(u)]]
        print('MULTI %-22s %2d sequences: %.0f scans/s; sequences that differ from the solo run: %s' % (os.path.basename(pipe), nseq, summ['steady_scans_per_s'], bad), flush=True)
PY
timeout 600 python bench.py --steps 10 --warmup 3 --no-extras 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('BENCH', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('parity_vs_cpu'))"
timeout 300 python tools/phase_probe.py 2>&1 | tail -9
