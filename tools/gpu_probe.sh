#!/bin/bash
cd /root/repo
run() {  # variant streams
  MH_MATCH=$1 timeout 300 python bench.py --streams $2 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('$1 S=$2: %.0f scans/s  k_match %.1f us  frac %.3f' % (d['value'], 1e3*d['roofline']['avg_kernel_ms'], d['roofline']['frac']))"
}
MH_MATCH=q timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_odometry.py -m gpu -x -q 2>&1 | tail -2
for S in 1 4 16; do run q $S; done
MH_MATCH=q python -m mola_lidar_odometry_amd.run_odometry --synthetic 40 2>&1 | tail -1
