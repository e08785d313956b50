python -m pytest tests/test_gpu_preprocess.py tests/test_odometry.py -m gpu -x -q 2>&1 | tail -2
python tools/fuzz_map_insert.py 120 777 2>&1 | tail -1
python tools/scratch/ms_probe.py 150 1,1 "" "MH_MAP_RADIX_SORT_NEW=1" 2>&1 | cut -c1-560 | tail -4
python tools/map_insert_time.py 2>&1 | tail -6 | head -5
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/po -o po -- env PYTHONPATH=$R python -m mola_lidar_odometry_amd.run_odometry --synthetic 40 --out-dir /tmp/odo > /dev/null 2>&1
python - <<PY
import csv,glob
for r in list(csv.DictReader(open(glob.glob("/tmp/po/**/po_kernel_stats.csv",recursive=True)[0]))):
    if "sort" in r["Name"] or "merge" in r["Name"] or "transform" in r["Name"]: print(r["Name"][60:200], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
