#!/bin/bash
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO
cp mola_lidar_odometry_amd/libmolahip.so /tmp/rel.so
timeout 300 python tools/phase_probe.py 2>&1 | tail -6
cp /tmp/rel.so mola_lidar_odometry_amd/libmolahip.so
