#!/bin/bash
# development probe: PMC counters for the match kernel (separate passes, no tracing flags besides kernel-trace)
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
OUT=$REPO/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $OUT/lds_counters.txt
i=0
# PMC_GROUPS: counter groups of one pass each, separated by ';' (default: the SQ view of rounds 1-4)
GROUPS_DEFAULT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
IFS=';' read -ra GROUPS_ARR <<< "${PMC_GROUPS:-$GROUPS_DEFAULT}"
for C in "${GROUPS_ARR[@]}"; do
  i=$((i+1))
  timeout ${PMC_TIMEOUT:-300} rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -o p -- python $REPO/bench.py --steps 2 --warmup 1 --streams ${PMC_STREAMS:-1} --no-cpu-baseline --no-profile --no-shared-run --no-extras --no-io $PMC_EXTRA > $OUT/p$i.log 2>&1
done
cd $REPO
cat $OUT/lds_counters.txt; echo
python - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
dur=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc/p*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:24]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob('gpurun_out/pmc/p*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        dur[r['Kernel_Name'][:24]].append(float(r['End_Timestamp'])-float(r['Start_Timestamp']))
for k in acc:
    if not (k.startswith('void k_match') or k.startswith('k_match') or k.startswith('k_accum')): continue
    print(k, 'avg duration us %.1f' % (sum(dur[k])/max(len(dur[k]),1)/1e3))
    for c,v in sorted(acc[k].items()):
        print('   %-36s n=%4d avg=%.4g'%(c,len(v),sum(v)/len(v)))
PY
tail -3 $OUT/p2.log
