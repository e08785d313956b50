#!/bin/bash
# scratch: A/B runs on the GPU box
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for w in c2 creal; do for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$w:', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'match us/scan', round(1e3*d['roofline']['avg_kernel_ms'],2))"; done; done
