#!/bin/bash
# development probe (rewritten per experiment; run on the GPU box through gpurun)
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 300 > gpurun_out/pytest_probe.log 2>&1
tail -4 gpurun_out/pytest_probe.log | cut -c1-400
ab() {
for r in 1 2; do
MH_MATCH=q timeout 600 python bench.py --no-cpu-baseline --no-shared-run --io none > gpurun_out/bench_$1.log 2>&1; python tools/bench_brief.py gpurun_out/bench_$1.log
done
timeout 600 python bench.py --no-cpu-baseline --no-shared-run --io none --workload creal > gpurun_out/bench_creal_$1.log 2>&1; python tools/bench_brief.py gpurun_out/bench_creal_$1.log
timeout 600 python bench.py --no-cpu-baseline --no-shared-run > gpurun_out/bench_io_$1.log 2>&1; python tools/bench_brief.py gpurun_out/bench_io_$1.log
}
ab new
cp mola_lidar_odometry_amd/libmolahip.so /tmp/new.so
cp tools/_ab/libmolahip_prev.so mola_lidar_odometry_amd/libmolahip.so
ab prev
cp /tmp/new.so mola_lidar_odometry_amd/libmolahip.so
