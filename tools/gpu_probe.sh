#!/bin/bash
# scratch: A/B runs on the GPU box
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
MH_MATCH=s timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -2
MH_MATCH=q timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_odometry.py -m gpu -x -q 2>&1 | tail -2
python -m mola_lidar_odometry_amd.run_odometry --synthetic 200 2>&1 | head -1 | cut -c1-120,330-640
MH_MATCH=q python -m mola_lidar_odometry_amd.run_odometry --synthetic 200 2>&1 | head -1 | cut -c1-120,330-640
