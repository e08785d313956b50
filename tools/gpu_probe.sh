#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for p in ndt; do
timeout 300 python -m mola_lidar_odometry_amd.run_odometry --synthetic 200 --pipeline pipelines/lidar3d-$p-hip.yaml 2>&1 | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['scans_per_s'], d['steady_scans_per_s'], d['startup_s_first_3_scans']); print(d['host_ms_per_scan'])"
done
