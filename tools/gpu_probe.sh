#!/bin/bash
# development probe (rewritten per experiment; run on the GPU box through gpurun)
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu --timeout 300 > gpurun_out/pytest_probe.log 2>&1
tail -4 gpurun_out/pytest_probe.log | cut -c1-400
for r in 1 2; do
MH_MATCH=q timeout 600 python bench.py --no-cpu-baseline --no-shared-run --io none > gpurun_out/bench_new.log 2>&1; python tools/bench_brief.py gpurun_out/bench_new.log
MH_NO_PREV_BOUND=1 MH_MATCH=q timeout 600 python bench.py --no-cpu-baseline --no-shared-run --io none > gpurun_out/bench_nob.log 2>&1; python tools/bench_brief.py gpurun_out/bench_nob.log
done
MH_MATCH=q timeout 600 python bench.py --no-cpu-baseline --no-shared-run --streams 1 --io none > gpurun_out/bench_s1.log 2>&1; python tools/bench_brief.py gpurun_out/bench_s1.log
