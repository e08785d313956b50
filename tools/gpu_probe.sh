#!/bin/bash
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; python tools/bench_brief.py gpurun_out/bench_default.log
timeout 900 python bench.py --no-cpu-baseline --no-shared-run --io none > gpurun_out/bench_res.log 2>&1; python tools/bench_brief.py gpurun_out/bench_res.log
timeout 900 python bench.py --no-cpu-baseline --no-shared-run --workload creal > gpurun_out/bench_creal_io.log 2>&1; python tools/bench_brief.py gpurun_out/bench_creal_io.log
timeout 900 python bench.py --no-cpu-baseline --no-shared-run --io none --streams 1 > gpurun_out/bench_s1.log 2>&1; python tools/bench_brief.py gpurun_out/bench_s1.log
timeout 1200 python tools/multi_seq_bench.py 200 1,2,4,8,16 2>&1 | tail -7 > gpurun_out/multi_seq.log; cat gpurun_out/multi_seq.log
timeout 1200 python tools/multi_seq_bench.py 120 4,8 pipelines/lidar3d-ndt-hip.yaml 2>&1 | tail -3 > gpurun_out/multi_seq_ndt.log; cat gpurun_out/multi_seq_ndt.log
