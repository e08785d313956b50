#!/bin/bash
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO
mkdir -p gpurun_out
cp mola_lidar_odometry_amd/libmolahip.so /tmp/cur.so
for v in 3_8_3 4_7_4 4_7_3; do
cp tools/_ab/lib_$v.so mola_lidar_odometry_amd/libmolahip.so
MH_MATCH=q timeout 600 python bench.py --no-cpu-baseline --no-shared-run --io none > gpurun_out/bench_$v.log 2>&1; echo -n "$v halves "; python tools/bench_brief.py gpurun_out/bench_$v.log
MH_NO_HALVES=1 MH_MATCH=q timeout 600 python bench.py --no-cpu-baseline --no-shared-run --io none > gpurun_out/bench_$v.log 2>&1; echo -n "$v nohalves "; python tools/bench_brief.py gpurun_out/bench_$v.log
done
cp /tmp/cur.so mola_lidar_odometry_amd/libmolahip.so
