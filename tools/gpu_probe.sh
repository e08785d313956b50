#!/bin/bash
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu --timeout 300 > gpurun_out/pytest_probe.log 2>&1
tail -3 gpurun_out/pytest_probe.log | cut -c1-400
timeout 900 python tools/multi_seq_bench.py 150 1,4 2>&1 | tail -3 | head -2
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/st2 -o st -- env PYTHONPATH=$REPO python -m mola_lidar_odometry_amd.run_odometry --synthetic 40 --out-dir $REPO/gpurun_out/odometry > /dev/null 2>&1
head -4 $REPO/gpurun_out/st2/*kernel_stats.csv | cut -c1-150
