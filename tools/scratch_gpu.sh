python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python - <<'PY'
import sys, os, json, subprocess, tempfile, time
sys.path.insert(0, '.')
import bench
from mola_lidar_odometry_amd import synth
ws, drive = bench.generate_inputs("small", [0], 60)
tmp = tempfile.mkdtemp(prefix="molahip_p_")
seq = synth.write_kitti_sequence(tmp, drive)
for env in ({}, {"MH_MAP_SIDE_STREAM": "0"}, {"MH_CHUNK_MARGIN": "6"}, {"MH_CHUNK_MARGIN": "0"}):
    os.environ.update(env)
    per, prof, _ = bench.run_lo_cli(seq, 1, os.path.join(tmp, "solo"))
    for k in env: os.environ.pop(k)
    print(env, "steady %.1f scans/s" % per[0]["steady_scans_per_s"], json.dumps({k: round(v, 4) for k, v in prof[0].items()}))
PY
