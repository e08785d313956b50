#!/bin/bash
# scratch: A/B runs on the GPU box
cd /root/repo
for np in 2 4; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $np --master-addr 127.0.0.1 --master-port 29511 -m mola_lidar_odometry_amd.run_odometry --synthetic 200 --copies $np 2>&1 | grep summary | cut -c1-200
done
timeout 400 python -m mola_lidar_odometry_amd.run_odometry --synthetic 200 --copies 2 2>&1 | grep summary | cut -c1-200
