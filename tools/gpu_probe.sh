#!/bin/bash
# development probe (rewritten per experiment; run on the GPU box through gpurun)
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO
mkdir -p gpurun_out
for m in o q; do
MH_MATCH=$m timeout 600 python bench.py --no-cpu-baseline --no-shared-run --io none > gpurun_out/bench_$m.log 2>&1; python tools/bench_brief.py gpurun_out/bench_$m.log
MH_MATCH=$m timeout 600 python bench.py --no-cpu-baseline --no-shared-run --streams 1 --io none > gpurun_out/bench_${m}_s1.log 2>&1; python tools/bench_brief.py gpurun_out/bench_${m}_s1.log
done
grep -o '"parity[^}]*}' gpurun_out/bench_o.log | head -3
