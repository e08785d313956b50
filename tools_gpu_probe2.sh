#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -12
for V in p 1 q8; do for S in 1 8; do
  (MH_MATCH=$V timeout 300 python bench.py --steps 5 --warmup 2 --streams $S --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/b.log
  python -c "
import json,sys
d=json.loads(open('gpurun_out/b.log').read().strip().splitlines()[-1])
print('V=$V S=$S', round(d['value'],1),'scans/s', 'ms/step', round(d['ms_per_step'],2), 'match avg ms', d['roofline'] and round(d['roofline']['avg_kernel_ms'],4), 'frac', d['roofline'] and round(d['roofline']['frac'],3))
"; done; done
