#!/bin/bash
cd /root/repo
cp mola_lidar_odometry_amd/libmolahip.so /tmp/prod.so
python tools/phase_probe.py
cp /tmp/prod.so mola_lidar_odometry_amd/libmolahip.so
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for p in default default ndt; do
python -m mola_lidar_odometry_amd.run_odometry --synthetic 200 --pipeline pipelines/lidar3d-$p-hip.yaml 2>&1 | head -1 | cut -c1-120,330-700
done
