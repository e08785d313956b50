#!/bin/bash
# kernel statistics of the NDT pipeline's odometry run (config-5 proxy), 300 scans of the city drive
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python -c "
import sys; sys.path.insert(0, '$REPO')
from mola_lidar_odometry_amd import synth_city
print(synth_city.write_kitti_drive('$OUT/city', 300, time_channel=True)[0])" > $OUT/ndt_city_dir.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/odom_ndt -o r04_odom_ndt -- $REPO/mola_lidar_odometry_amd/molahip-lo-cli --pipeline $REPO/pipelines/lidar3d-ndt-hip.yaml --seq-dir $(cat $OUT/ndt_city_dir.txt) --time-field 12 --profile --out $OUT/r04_odom_ndt.tum > $OUT/r04_odom_ndt_stdout.log 2>&1
$REPO/mola_lidar_odometry_amd/molahip-lo-cli --pipeline $REPO/pipelines/lidar3d-ndt-hip.yaml --seq-dir $(cat $OUT/ndt_city_dir.txt) --time-field 12 --profile --out $OUT/r04_odom_ndt2.tum > $OUT/r04_odom_ndt_noprof_stdout.log 2>&1
rm -rf $OUT/city
find $OUT/odom_ndt -name "*kernel_stats.csv" -exec cp {} $OUT/r04_odom_ndt_kernel_stats.csv \;
head -30 $OUT/r04_odom_ndt_kernel_stats.csv | cut -c1-60,200-
tail -20 $OUT/r04_odom_ndt_noprof_stdout.log
