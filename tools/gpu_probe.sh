#!/bin/bash
# scratch: A/B runs on the GPU box
cd /root/repo
run() {  # label streams
  timeout 300 python bench.py --streams $2 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('$1 S=$2: %.0f scans/s  k_match %.1f us  frac %.3f' % (d['value'], 1e3*d['roofline']['avg_kernel_ms'], d['roofline']['frac']))"
}
run default 16
run default 32
GPU_MAX_HW_QUEUES=8 run q8 16
GPU_MAX_HW_QUEUES=16 run q16 16
GPU_MAX_HW_QUEUES=16 run q16 32
GPU_MAX_HW_QUEUES=2 run q2 16
MH_NO_GRAPH=1 run nograph 16
MH_NO_GRAPH=1 GPU_MAX_HW_QUEUES=16 run nograph_q16 16
