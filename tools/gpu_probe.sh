#!/bin/bash
# development probe (rewritten per experiment; run on the GPU box through gpurun)
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "batch or lockstep" > gpurun_out/pytest_batch.log 2>&1
tail -5 gpurun_out/pytest_batch.log
timeout 600 python bench.py > gpurun_out/bench_default.log 2>&1; tail -c 6000 gpurun_out/bench_default.log
timeout 300 python bench.py --no-io --no-cpu-baseline --no-shared-run > gpurun_out/bench_noio.log 2>&1; tail -c 1500 gpurun_out/bench_noio.log
timeout 300 python bench.py --maps shared --no-cpu-baseline > gpurun_out/bench_shared.log 2>&1; tail -c 1500 gpurun_out/bench_shared.log
