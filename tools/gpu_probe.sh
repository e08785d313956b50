#!/bin/bash
# scratch: A/B runs on the GPU box
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/row_vs_quad.py 2>&1 | tail -12
for p in default; do
timeout 300 python -m mola_lidar_odometry_amd.run_odometry --synthetic 200 --pipeline pipelines/lidar3d-$p-hip.yaml 2>&1 | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['scans_per_s'], d['steady_scans_per_s'], d['host_ms_per_scan']['onLidar.3.run_icp'])"
done
