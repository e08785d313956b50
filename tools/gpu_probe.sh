#!/bin/bash
# scratch: A/B runs on the GPU box
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 -m mola_lidar_odometry_amd.run_odometry --synthetic 120 --copies 2 2>&1 | grep summary | cut -c1-200
timeout 600 python bench.py 2>&1 | tail -1 | cut -c1-250
