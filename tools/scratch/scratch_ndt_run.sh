#!/bin/bash
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
N=${1:-400}
python -c "
import sys; sys.path.insert(0, '$REPO')
from mola_lidar_odometry_amd import synth_city
print(synth_city.write_kitti_drive('/tmp/city', $N, time_channel=True)[0])" > /tmp/city_dir.txt
for P in lidar3d-ndt-hip lidar3d-default-hip; do
$REPO/mola_lidar_odometry_amd/molahip-lo-cli --pipeline $REPO/pipelines/$P.yaml --seq-dir $(cat /tmp/city_dir.txt) --time-field 12 --profile --out $OUT/run_$P.tum > $OUT/run_$P.log 2>&1
tail -2 $OUT/run_$P.log | cut -c1-1200
md5sum $OUT/run_$P.tum
done
