#!/bin/bash
# scratch: A/B runs on the GPU box
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for e in "" "MH_NO_LOCKSTEP=1"; do env $e timeout 300 python bench.py --no-cpu-baseline --workload creal 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('creal $e:', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'match us/scan', round(1e3*d['roofline']['avg_kernel_ms'],2) if d['roofline'] else None)"; done
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-140
