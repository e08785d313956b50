#!/bin/bash
# FETCH_SIZE calibration (tools/fetch_calib.hip) -> gpurun_out/fetch_calib.json ; run on the GPU box through gpurun
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
OUT=$REPO/gpurun_out/fetch_calib
rm -rf $OUT; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $REPO/tools/fetch_calib.hip -o /tmp/fetch_calib || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_RD[A-Za-z0-9_]*\|TCC_EA0_WR[A-Za-z0-9_]*\|TCC_HIT[A-Za-z0-9_]*\|TCC_MISS[A-Za-z0-9_]*" | sort -u | tr '\n' ' ' > $OUT/tcc_counters.txt
i=0
for C in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -o p -- /tmp/fetch_calib > $OUT/p$i.log 2>&1
done
cd $REPO
python - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/fetch_calib/p*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
info = json.loads([l for l in open('gpurun_out/fetch_calib/p1.log') if l.startswith('{')][-1])
req = info['requested_bytes_per_launch']
res = {'requested_bytes_per_launch': req, 'table_bytes': info['table_bytes'], 'kernels': {},
       'available_tcc_counters': open('gpurun_out/fetch_calib/tcc_counters.txt').read().split()}
for k, v in sorted(acc.items()):
    if 'k_calib' not in k and 'k_stream' not in k: continue
    row = {c: sum(x) / len(x) for c, x in v.items()}
    if 'FETCH_SIZE' in row:
        row['FETCH_SIZE_bytes'] = row['FETCH_SIZE'] * 1024.0
        row['FETCH_SIZE_bytes_over_requested'] = row['FETCH_SIZE_bytes'] / req
    res['kernels'][k] = row
json.dump(res, open('gpurun_out/fetch_calib.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
