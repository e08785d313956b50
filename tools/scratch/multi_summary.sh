#!/bin/bash
# summary lines (batcher statistics) of N copies of the city drive through one molahip-lo-cli process
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)
N=${1:-400}
if [ ! -f /tmp/city_dir.txt ]; then
python -c "
import sys; sys.path.insert(0, '$REPO')
from mola_lidar_odometry_amd import synth_city
print(synth_city.write_kitti_drive('/tmp/city', $N, time_channel=True)[0])" > /tmp/city_dir.txt
fi
D=$(cat /tmp/city_dir.txt)
for S in ${2:-8 16}; do
  ARGS=""
  for i in $(seq $S); do ARGS="$ARGS --seq-dir $D"; done
  $REPO/mola_lidar_odometry_amd/molahip-lo-cli --pipeline $REPO/pipelines/${3:-lidar3d-default-hip}.yaml $ARGS --time-field 12 --profile --out /tmp/ms_$S.tum 2>&1 | grep -E "\"sequences\"" | cut -c1-700
done
