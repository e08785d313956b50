#!/bin/bash
# development probe (rewritten per experiment; run on the GPU box through gpurun)
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; tail -4 gpurun_out/pytest_gpu.log | cut -c1-300
timeout 600 python bench.py > gpurun_out/bench_default.log 2>&1; python tools/bench_brief.py gpurun_out/bench_default.log
timeout 600 python bench.py --io none --no-cpu-baseline --no-shared-run > gpurun_out/bench_io_none.log 2>&1; python tools/bench_brief.py gpurun_out/bench_io_none.log
timeout 600 python bench.py --workload creal --no-cpu-baseline --no-shared-run > gpurun_out/bench_creal.log 2>&1; python tools/bench_brief.py gpurun_out/bench_creal.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
