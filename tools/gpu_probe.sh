#!/bin/bash
# development probe (rewritten per experiment; run on the GPU box through gpurun)
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO
mkdir -p gpurun_out
timeout 1200 python tools/multi_seq_bench.py 200 1,2,4,8,16 > gpurun_out/multi_seq.log 2>&1; tail -7 gpurun_out/multi_seq.log | cut -c1-400
PYTHONPATH=$REPO timeout 600 python -m mola_lidar_odometry_amd.run_odometry --synthetic 200 --out-dir gpurun_out/odometry > gpurun_out/odom200.log 2>&1; tail -3 gpurun_out/odom200.log | cut -c1-1500
