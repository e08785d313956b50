#!/bin/bash
# scratch: A/B runs on the GPU box
cd /root/repo
for s in 8 16 32 64 96; do timeout 300 python bench.py --no-cpu-baseline --streams $s 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('S=$s:', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'match us/scan', round(1e3*d['roofline']['avg_kernel_ms'],2))"; done
