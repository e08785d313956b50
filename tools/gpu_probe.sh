#!/bin/bash
# scratch: A/B runs on the GPU box
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for a in "" "--no-prefetch"; do
for p in default ndt; do
timeout 300 python -m mola_lidar_odometry_amd.run_odometry --synthetic 200 --pipeline pipelines/lidar3d-$p-hip.yaml $a 2>&1 | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$p $a', round(d['scans_per_s'],1), 'steady', round(d['steady_scans_per_s'],1), 'ate', round(d['ate_rmse_m'],4)); print({k:v for k,v in d['host_ms_per_scan'].items() if v>0.02})"
done
done
