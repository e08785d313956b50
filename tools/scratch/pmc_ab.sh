#!/bin/bash
# HBM-side bytes of k_match4_b (FETCH_SIZE / WRITE_SIZE passes) for the library in MOLAHIP_LIB_PATH (or the product)
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)
TAG=$1
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | cut -c1-5)
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_${TAG}_$N -o x -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-shared-run --no-extras --upload-thread 0 > /tmp/pmc_${TAG}_$N.log 2>&1
  python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('/tmp/pmc_${TAG}_$N/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith('k_match4_b'): acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print('$TAG', k, sum(v)/len(v), len(v))
PY
done
