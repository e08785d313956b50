#!/bin/bash
# development probe: PMC counters for the match kernel (separate passes, no tracing flags besides kernel-trace)
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
V=${MH_MATCH:-q}
i=0
for C in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM" \
         "TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  MH_MATCH=$V timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /root/repo/gpurun_out/pmc/p$i -o p -- python /root/repo/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-profile > /root/repo/gpurun_out/pmc/p$i.log 2>&1
done
cd /root/repo
python - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc/p*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:24]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in acc:
    if not (k.startswith('void k_match') or k.startswith('k_match')): continue
    print(k)
    for c,v in sorted(acc[k].items()):
        print('   %-36s n=%4d avg=%.4g'%(c,len(v),sum(v)/len(v)))
PY
