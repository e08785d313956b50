#!/bin/bash
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for V in p; do for S in 1 8 16; do MH_MATCH=$V timeout 300 python bench.py --steps 5 --warmup 2 --streams $S --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$V S=$S', round(d['value'],1),'scans/s', 'match avg ms', round(d['roofline']['avg_kernel_ms'],4), 'frac', round(d['roofline']['frac'],3))"; done; done
