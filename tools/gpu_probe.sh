#!/bin/bash
# scratch: A/B runs on the GPU box
cd /root/repo
MH_MATCH=q timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
for S in 1 32; do python bench.py --no-cpu-baseline --streams $S --steps 10 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('S=$S: %.0f scans/s  k_match %.1f us' % (d['value'], 1e3*d['roofline']['avg_kernel_ms']))"; done
