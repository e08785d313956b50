#!/bin/bash
# scratch probe used during development (not part of the product)
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1
tail -3 gpurun_out/pytest_gpu.log; grep -E "^E  |^FAILED|Error" gpurun_out/pytest_gpu.log | head -20
for V in 1 q4 q8 q16; do for S in 1 8; do
  (MH_MATCH=$V timeout 300 python bench.py --steps 5 --warmup 2 --streams $S --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_$V_s$S.log
  python -c "
import json,sys
d=json.loads(open('gpurun_out/bench_$V_s$S.log').read().strip().splitlines()[-1])
print('V=$V S=$S', round(d['value'],1),'scans/s', 'ms/step', round(d['ms_per_step'],2), 'match avg ms', d['roofline'] and round(d['roofline']['avg_kernel_ms'],4), 'frac', d['roofline'] and round(d['roofline']['frac'],3))
"; done; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof1 -o s1 -- python /root/repo/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-profile > /root/repo/gpurun_out/prof1.log 2>&1
cd /root/repo
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof1/s1_kernel_stats.csv')))
for r in rows[:6]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(5), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(9),'us', r['Percentage'])
PY
