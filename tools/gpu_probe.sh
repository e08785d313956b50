#!/bin/bash
# development probe (rewritten per experiment; run on the GPU box through gpurun)
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu --timeout 300 > gpurun_out/pytest_probe.log 2>&1
tail -4 gpurun_out/pytest_probe.log | cut -c1-400
for s in 16 32 48 64; do
timeout 600 python bench.py --no-cpu-baseline --no-shared-run --io none --streams $s > gpurun_out/bench_s$s.log 2>&1; python tools/bench_brief.py gpurun_out/bench_s$s.log
done
timeout 600 python bench.py --no-cpu-baseline --no-shared-run > gpurun_out/bench_io.log 2>&1; python tools/bench_brief.py gpurun_out/bench_io.log
timeout 600 python bench.py --no-cpu-baseline --no-shared-run --io none --workload creal > gpurun_out/bench_creal.log 2>&1; python tools/bench_brief.py gpurun_out/bench_creal.log
