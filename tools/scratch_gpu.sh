python -m pytest tests -m gpu -x -q 2>&1 | tail -5
bash tools/fetch_calib.sh 2>&1 | tail -60
python - <<'PY'
import sys, os, json, tempfile
sys.path.insert(0, '.')
import bench
from mola_lidar_odometry_amd import synth
ws, drive = bench.generate_inputs("small", [0], 100)
tmp = tempfile.mkdtemp(prefix="molahip_p_")
seq = synth.write_kitti_sequence(tmp, drive)
for n in (1, 4, 8, 16):
    per, prof, summ = bench.run_lo_cli(seq, n, os.path.join(tmp, "m%d" % n))
    print("N=%d" % n, "steady", (summ or per[0])["steady_scans_per_s"], "whole", (summ or per[0])["scans_per_s"])
    print("   seq0 profile:", json.dumps({k: round(v, 4) for k, v in prof[0].items()}))
PY
