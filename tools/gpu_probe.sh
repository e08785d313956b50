#!/bin/bash
# scratch: A/B runs on the GPU box
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python -m mola_lidar_odometry_amd.run_odometry --synthetic 200 2>&1 | head -1 | cut -c1-120,330-640
MH_NO_ONE_GROUP=1 python -m mola_lidar_odometry_amd.run_odometry --synthetic 200 2>&1 | head -1 | cut -c1-120,330-640
python tools/row_vs_quad.py 2>&1 | grep "^n=" | head -4
