#!/bin/bash
# scratch: A/B runs on the GPU box
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python -m mola_lidar_odometry_amd.run_odometry --synthetic 200 2>&1 | head -1 | cut -c1-120,330-640
for S in 1 32; do python bench.py --no-cpu-baseline --streams $S --steps 10 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('S=$S: %.0f scans/s  k_match %.1f us' % (d['value'], 1e3*d['roofline']['avg_kernel_ms']))"; done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o x -- python /root/repo/bench.py --no-cpu-baseline --streams 1 --steps 5 > /dev/null 2>&1; python - <<'PY'
import csv, glob
for r in list(csv.DictReader(open(glob.glob('/tmp/pp/*kernel_stats.csv')[0])))[:4]:
    print(r['Name'][:40].ljust(40), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(8),'us')
PY
