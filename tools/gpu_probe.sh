#!/bin/bash
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu --timeout 300 > gpurun_out/pytest_probe.log 2>&1
tail -3 gpurun_out/pytest_probe.log | cut -c1-400
for e in "" "MH_NO_PREV_BOUND=1"; do
echo "== $e"
env $e timeout 600 python bench.py --no-cpu-baseline --no-shared-run --io none --workload creal > gpurun_out/bench_creal.log 2>&1; python tools/bench_brief.py gpurun_out/bench_creal.log
env $e timeout 900 python tools/multi_seq_bench.py 120 1,8 2>&1 | tail -4
done
