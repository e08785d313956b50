python -m pytest tests/test_gpu_preprocess.py tests/test_odometry.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
python tools/fuzz_map_insert.py 200 31337 2>&1 | tail -1
python tools/fuzz_odometry.py 10 4242 2>&1 | tail -1
python tools/scratch/ms_probe.py 150 1,1 "" "MH_MAP_NO_COLLECT=1" 2>&1 | cut -c1-560 | tail -4
python tools/map_insert_time.py 2>&1 | tail -6 | head -4
