#!/bin/bash
# the NDT pipeline over the city drive, several times: failures of the device loop and the trajectories' checksums
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)
N=${1:-150}; R=${2:-8}
if [ ! -f /tmp/city_dir.txt ]; then
python -c "
import sys; sys.path.insert(0, '$REPO')
from mola_lidar_odometry_amd import synth_city
print(synth_city.write_kitti_drive('/tmp/city', $N, time_channel=True)[0])" > /tmp/city_dir.txt
fi
for i in $(seq $R); do
  $REPO/mola_lidar_odometry_amd/molahip-lo-cli --pipeline $REPO/pipelines/${3:-lidar3d-ndt-hip}.yaml --seq-dir $(cat /tmp/city_dir.txt) --time-field 12 --out /tmp/loop_$i.tum > /tmp/loop_$i.log 2>&1
  echo "run $i rc=$? $(md5sum < /tmp/loop_$i.tum | cut -c1-8) $(grep -o 'gave up.*' /tmp/loop_$i.log | head -1 | cut -c60-400)"
done
