#!/bin/bash
cd /root/repo
cp mola_lidar_odometry_amd/libmolahip.so /tmp/prod.so
python tools/phase_probe.py 2>&1 | tail -3
cp /tmp/prod.so mola_lidar_odometry_amd/libmolahip.so
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
MH_NO_GRAPH=1 timeout 300 python tools/og_sweep.py 2>&1 | head -4
for p in default ndt; do
timeout 300 python -m mola_lidar_odometry_amd.run_odometry --synthetic 200 --pipeline pipelines/lidar3d-$p-hip.yaml 2>&1 | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['scans_per_s'], d['steady_scans_per_s'], d['host_ms_per_scan']['onLidar.3.run_icp'])"
done
python bench.py --no-cpu-baseline --streams 1 2>&1 | tail -1 | cut -c1-200
