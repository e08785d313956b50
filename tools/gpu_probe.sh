#!/bin/bash
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO
mkdir -p gpurun_out
cp mola_lidar_odometry_amd/libmolahip.so /tmp/cur.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu --timeout 300 > gpurun_out/pytest_probe.log 2>&1
tail -3 gpurun_out/pytest_probe.log | cut -c1-400
for v in cur a6 a7 a8 cur; do
if [ $v = cur ]; then cp /tmp/cur.so mola_lidar_odometry_amd/libmolahip.so; else cp tools/_ab/lib_$v.so mola_lidar_odometry_amd/libmolahip.so; fi
MH_MATCH=q timeout 600 python bench.py --no-cpu-baseline --no-shared-run --io none > gpurun_out/bench_$v.log 2>&1; echo -n "$v "; python tools/bench_brief.py gpurun_out/bench_$v.log
done
cp /tmp/cur.so mola_lidar_odometry_amd/libmolahip.so
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/st -o st -- python $REPO/bench.py --no-cpu-baseline --no-shared-run --io none --upload-thread 0 > /dev/null 2>&1
head -6 $REPO/gpurun_out/st/*kernel_stats.csv | cut -c1-120
