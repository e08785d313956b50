#!/bin/bash
# development probe (rewritten per experiment; run on the GPU box through gpurun)
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO
mkdir -p gpurun_out
for d in 0 1 2 4; do
timeout 600 python bench.py --no-cpu-baseline --no-shared-run --io both --upload-delay-ms $d > gpurun_out/bench_d$d.log 2>&1; python tools/bench_brief.py gpurun_out/bench_d$d.log
done
timeout 600 python bench.py --no-cpu-baseline --no-shared-run --io none > gpurun_out/bench_io_none.log 2>&1; python tools/bench_brief.py gpurun_out/bench_io_none.log
