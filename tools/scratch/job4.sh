timeout 300 python -m pytest tests/test_odometry.py tests/test_host_layer.py -m gpu -x -q 2>&1 | tail -3
python tools/scratch/ms_probe.py 150 8,8,4,16 "" 2>&1 | cut -c1-330 | tail -4
timeout 600 python tools/stress_cli_sequences.py 3 2>&1 | tail -14
