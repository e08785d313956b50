set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python - <<'PY'
import os, sys, json, subprocess, tempfile
sys.path.insert(0, os.getcwd())
import bench
from mola_lidar_odometry_amd import synth_city
tmp = tempfile.mkdtemp(prefix='molahip_ms_')
seq, drive = synth_city.write_kitti_drive(tmp, 400, time_channel=True)
solo = None
for name, env in (('streaming', {}), ('chunks', {'MH_NO_STREAM': '1'})):
    for pipe in (bench.PIPELINE, bench.PIPELINE_NDT):
        for c in (1, 4, 8, 16):
            per, prof, summ = bench.run_lo_cli(seq, c, os.path.join(tmp, 'o_%s_%d' % (name, c)), pipeline=pipe, env=env)
            s = summ or per[0]
            extra = ('batch run %.3f ms, assemble %.3f ms, jobs/batch %.1f' % (summ['ms_per_batch_running'], summ['ms_per_batch_assembling'], summ['jobs_per_batch'])) if summ else ''
            print('MS %-10s %-22s %2d seq: steady %.0f scans/s whole %.0f  %s' % (name, os.path.basename(pipe), c, s['steady_scans_per_s'], s['scans_per_s'], extra), flush=True)
            if pipe == bench.PIPELINE:
                t = open(per[0]['tum']).read()
                solo = solo or t
                assert all(open(q['tum']).read() == solo for q in per), 'trajectory differs'
print('all default-pipeline trajectories identical to the first solo run')
PY
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench3.json 2> gpurun_out/r04/bench3.err; tail -8 gpurun_out/r04/bench3.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04/bench3.json'))
print('BENCH', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_of_floor'), d['roofline']['avg_launch_ms'])
s=d['single_sequence']; print('SS', s['value'], s['whole_run_scans_per_s'], s['ratio_vs_cpu_driver'], s['cpu_driver'].get('value_c_library_only'), s['cpu_driver'].get('cores'), s.get('with_1M_point_local_map'))
n=d['single_sequence_ndt']; print('NDT', n['value'], n['ratio_vs_cpu_driver'], n['cpu_driver'].get('value_c_library_only'))
print('MULTI', d['multi_sequence'])
PY
