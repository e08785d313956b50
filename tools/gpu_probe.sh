#!/bin/bash
# scratch: A/B runs on the GPU box
cd /root/repo
timeout 300 python -m mola_lidar_odometry_amd.run_odometry --synthetic 200 --copies 3 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if not l.startswith('{'): print(l.strip()[:200]); continue
    d=json.loads(l)
    if d.get('summary'): print(d)
    else: print({k:d[k] for k in ('sequence','scans','good','keyframes','icp_iterations','map_points','steady_scans_per_s','ate_rmse_m') if k in d})"
