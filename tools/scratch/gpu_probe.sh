#!/bin/bash
# development probe (edited per experiment; run on the GPU box through gpurun): the GPU suite and one bench line
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)
cd $REPO
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu --timeout 300 > gpurun_out/pytest_probe.log 2>&1
tail -3 gpurun_out/pytest_probe.log | cut -c1-400
timeout 900 python bench.py --no-cpu-baseline --no-shared-run > gpurun_out/bench_probe.log 2>&1; python tools/bench_brief.py gpurun_out/bench_probe.log
