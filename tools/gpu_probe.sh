#!/bin/bash
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO
mkdir -p gpurun_out
for io in none both none both; do
timeout 600 python bench.py --no-cpu-baseline --no-shared-run --io $io > gpurun_out/bench_io_$io.log 2>&1; echo -n "$io "; python tools/bench_brief.py gpurun_out/bench_io_$io.log | cut -c1-330
done
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; python tools/bench_brief.py gpurun_out/bench_default.log
tail -3 gpurun_out/bench_default.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 300 -k "batch" 2>&1 | tail -2
