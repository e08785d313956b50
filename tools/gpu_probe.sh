#!/bin/bash
# scratch: A/B runs on the GPU box
cd /root/repo
timeout 400 python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
from mola_lidar_odometry_amd import synth
d = synth.make_drive(200, rings=64, azimuths=1875)
os.makedirs("/tmp/kt/sequences/00/velodyne", exist_ok=True)
for k, (xyz, _) in enumerate(d["scans"]):
    np.concatenate([xyz, np.zeros((len(xyz), 1), np.float32)], 1).astype(np.float32).tofile("/tmp/kt/sequences/00/velodyne/%06d.bin" % k)
np.savetxt("/tmp/kt/sequences/00/times.txt", d["stamps"] - d["stamps"][0], fmt="%.6e")
print("tree written")
PY
for a in "" "--no-prefetch" ""; do
timeout 120 mola_lidar_odometry_amd/molahip-lo-cli --pipeline pipelines/lidar3d-default-hip.yaml --seq-dir /tmp/kt/sequences/00 --out /tmp/kt/00.tum $a | cut -c1-300
done
timeout 120 mola_lidar_odometry_amd/molahip-lo-cli --pipeline pipelines/lidar3d-ndt-hip.yaml --seq-dir /tmp/kt/sequences/00 --out /tmp/kt/00n.tum | cut -c1-300
