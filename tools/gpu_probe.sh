#!/bin/bash
# scratch: A/B runs on the GPU box
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python -m mola_lidar_odometry_amd.run_odometry --synthetic 200 2>&1 | head -1 | cut -c1-120,330-640
for S in 1 32; do python bench.py --no-cpu-baseline --streams $S --steps 10 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('S=$S: %.0f scans/s  k_match %.1f us' % (d['value'], 1e3*d['roofline']['avg_kernel_ms']))"; done
