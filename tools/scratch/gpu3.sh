set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_cli_devices.py tests/test_host_layer.py -m gpu -x -q 2>&1 | tail -8
timeout 1200 python tools/match_floor.py --out gpurun_out/r04/match_floor.json 2>&1 | tail -16
B="--steps 8 --warmup 3 --no-extras --no-cpu-baseline --no-shared-run"
for v in product carry4 carry3 carry4w7; do
  if [ $v = product ]; then unset MOLAHIP_LIB_PATH; else export MOLAHIP_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/libmolahip_$v.so; fi
  timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -2
  timeout 300 python bench.py $B > gpurun_out/r04/ab_$v.json 2> gpurun_out/r04/ab_$v.err
  python -c "
import json,sys
d=json.load(open('gpurun_out/r04/ab_$v.json')); print('AB $v', round(d['value'],1), 'launch_ms', round(d['roofline']['avg_launch_ms'],4), d.get('parity_vs_cpu'))"
done
unset MOLAHIP_LIB_PATH
MH_MATCH=o timeout 300 python bench.py $B > gpurun_out/r04/ab_sorted.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r04/ab_sorted.json')); print('AB MH_MATCH=o', round(d['value'],1), round(d['roofline']['avg_launch_ms'],4))"
MH_LOCKSTEP_GROUPS=2 timeout 300 python bench.py $B > gpurun_out/r04/ab_groups2.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r04/ab_groups2.json')); print('AB groups=2', round(d['value'],1), round(d['roofline']['avg_launch_ms'],4))"
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04/pmcc
mkdir -p $OUT
i=0
for C in "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32" \
         "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --scan-sets 1 --no-cpu-baseline --no-profile --no-shared-run --no-extras --upload-thread 0 > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, json
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/r04/pmcc/p*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
out={k:{c:sum(x)/len(x) for c,x in v.items()} for k,v in acc.items() if any(s in k for s in ('k_match4_b','k_accum_b'))}
json.dump(out, open('gpurun_out/r04/pmc_valu_mix.json','w'), indent=1); print(json.dumps(out, indent=1))
PY
