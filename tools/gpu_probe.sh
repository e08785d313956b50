#!/bin/bash
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO
mkdir -p gpurun_out
cp mola_lidar_odometry_amd/libmolahip.so /tmp/cur.so
for v in r1 r2 r4 r1 r2 r4; do
cp tools/_ab/lib_$v.so mola_lidar_odometry_amd/libmolahip.so
MH_MATCH=q timeout 600 python bench.py --no-cpu-baseline --no-shared-run --io none > gpurun_out/bench_$v.log 2>&1; echo -n "$v "; python tools/bench_brief.py gpurun_out/bench_$v.log
done
cp /tmp/cur.so mola_lidar_odometry_amd/libmolahip.so
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu --timeout 300 > gpurun_out/pytest_probe.log 2>&1
tail -2 gpurun_out/pytest_probe.log | cut -c1-300
