#!/bin/bash
# scratch: A/B runs on the GPU box
cd /root/repo
for q in 4 8 16 32; do
GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --no-cpu-baseline --workload creal --streams 32 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('creal queues=$q S=32', round(d['value'],1), 'scans/s  match us', round(1e3*d['roofline']['avg_kernel_ms'],1))"
done
