#!/bin/bash
cd /root/repo
for f in "" 1; do
env ${f:+MH_NO_FUSE16=1} timeout 300 python -m mola_lidar_odometry_amd.run_odometry --synthetic 120 --pipeline tools/_dense_icp_layer.yaml 2>&1 | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('nofuse=$f steady', round(d['steady_scans_per_s'],1), 'icp layer', d['mean_points_for_icp'], 'iters/scan', d['icp_iterations']/d['scans'], 'map', d['map_points'], 'ate', d.get('ate_rmse_m')); print({k:v for k,v in d['host_ms_per_scan'].items() if v>0.02})"
done
