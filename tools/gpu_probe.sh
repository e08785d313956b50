#!/bin/bash
# development probe (rewritten per experiment; run on the GPU box through gpurun)
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu --timeout 300 > gpurun_out/pytest_probe.log 2>&1
tail -3 gpurun_out/pytest_probe.log | cut -c1-400
for e in "" "MH_MAP_FULL_SORT=1"; do
echo "== $e"
env $e timeout 600 python tools/odom_profile.py 150 2>&1 | grep -E "steady|update_local_map|run_icp"
env $e timeout 900 python tools/multi_seq_bench.py 120 1,8 2>&1 | grep "sequences in one"
done
