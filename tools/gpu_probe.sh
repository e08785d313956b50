#!/bin/bash
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO
mkdir -p gpurun_out
cp mola_lidar_odometry_amd/libmolahip.so /tmp/rel.so
cp tools/libmolahip_dbg.so mola_lidar_odometry_amd/libmolahip.so
for st in 0 4 0 4; do
MH_DBG_STOP=$st timeout 600 python bench.py --no-cpu-baseline --no-shared-run --io none > gpurun_out/bench_dbg$st.log 2>&1; echo -n "stop=$st "; python tools/bench_brief.py gpurun_out/bench_dbg$st.log
done
cp /tmp/rel.so mola_lidar_odometry_amd/libmolahip.so
