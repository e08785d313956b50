set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
B="--steps 8 --warmup 3 --no-extras --no-cpu-baseline --no-shared-run"
for v in product wide4 narrow3 narrow4xyz; do
  if [ $v = product ]; then unset MOLAHIP_LIB_PATH; else export MOLAHIP_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/libmolahip_$v.so; fi
  timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -1
  for rep in 1 2; do
  timeout 300 python bench.py $B > gpurun_out/r04/ab2_$v.json 2> gpurun_out/r04/ab2_$v.err
  python -c "
import json
d=json.load(open('gpurun_out/r04/ab2_$v.json')); print('AB2 $v', round(d['value'],1), 'launch_ms', round(d['roofline']['avg_launch_ms'],4), 'parity', d['parity_vs_cpu']['downloaded_final_pairings_bit_equal'] if d.get('parity_vs_cpu') else None)"
  done
done
unset MOLAHIP_LIB_PATH
timeout 1200 python tools/match_floor.py --out gpurun_out/r04/match_floor.json 2>&1 | tail -18
# single sequence: streaming loop control vs round 3's chunks
python - <<'PY'
import os, sys, json, subprocess, tempfile
sys.path.insert(0, os.getcwd())
import bench
from mola_lidar_odometry_amd import synth_city
tmp = tempfile.mkdtemp(prefix='molahip_st_')
seq, drive = synth_city.write_kitti_drive(tmp, 600, time_channel=True)
for name, env in (('streaming', {}), ('chunks (MH_NO_STREAM=1)', {'MH_NO_STREAM': '1'}), ('streaming lead 3', {'MH_STREAM_LEAD': '3'}), ('streaming lead 1', {'MH_STREAM_LEAD': '1'})):
    for pipe in (bench.PIPELINE, bench.PIPELINE_NDT):
        per, prof, _ = bench.run_lo_cli(seq, 1, os.path.join(tmp, 'o'), pipeline=pipe, env=env)
        p = prof[0]
        print('STREAM %-26s %-22s steady %.0f scans/s  onLidar %.4f ms  icp %.4f  enq %.1f exec %.1f polls %.2f' % (name, os.path.basename(pipe), per[0]['steady_scans_per_s'], p['onLidar'], p['onLidar.3.run_icp'], p['icp.enqueued_iterations'], p['icp.executed_iterations'], p['icp.host_polls']), flush=True)
    open(os.path.join(tmp, name.split()[0] + '.tum'), 'w').write(open(per[0]['tum']).read())
print('identical trajectories:', open(os.path.join(tmp, 'streaming.tum')).read() == open(os.path.join(tmp, 'chunks.tum')).read())
PY
