cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, json, subprocess, tempfile
sys.path.insert(0, os.getcwd())
import bench
from mola_lidar_odometry_amd import synth_city
tmp = tempfile.mkdtemp(prefix='molahip_st_')
seq, drive = synth_city.write_kitti_drive(tmp, 300, time_channel=True)
for ename, env in (('MH_NO_STEP_CHAIN', {'MH_NO_STEP_CHAIN': '1'}), ('MH_NO_STEP_CHAIN + verify', {'MH_NO_STEP_CHAIN': '1', 'MH_DEBUG_VERIFY_BATCH': '1'})):
  for pipe in (bench.PIPELINE, bench.PIPELINE_NDT):
    for rep in range(4):
        per, prof, summ = bench.run_lo_cli(seq, 16, os.path.join(tmp, 'm'), pipeline=pipe, max_scans=300, env=env)
        txt = [open(q['tum']).read().splitlines() for q in per]
        ref = max(set(map(tuple, txt)), key=lambda t: sum(1 for u in txt if tuple(u) == t))
        bad = [(k, next(i for i, (a, b) in enumerate(zip(u, ref)) if a != b)) for k, u in enumerate(txt) if tuple(u) != ref]
        print('MULTI %-26s %-22s 16 sequences: %.0f scans/s; differ from the majority: %s' % (ename, os.path.basename(pipe), summ['steady_scans_per_s'], bad), flush=True)
PY
