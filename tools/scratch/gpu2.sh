set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04/gputests2.log; cat gpurun_out/r04/gputests2.log
timeout 900 python tools/match_floor.py --out gpurun_out/r04/match_floor.json 2>&1 | tail -12
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench2.json 2> gpurun_out/r04/bench2.err; tail -6 gpurun_out/r04/bench2.err
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04/pmcb
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $OUT/sq_counters.txt
rocprofv3 -L 2>/dev/null | grep -io "[A-Za-z]*Busy[A-Za-z]*\|[A-Za-z]*Util[A-Za-z]*" | sort -u | tr '\n' ' ' > $OUT/derived.txt
i=0
for C in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" \
         "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" \
         "VALUBusy" "SALUBusy" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --scan-sets 1 --no-cpu-baseline --no-profile --no-shared-run --no-extras --upload-thread 0 > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"; tail -2 $OUT/p$i.log | cut -c1-300
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, json
acc=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
for f in glob.glob('gpurun_out/r04/pmcb/p*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob('gpurun_out/r04/pmcb/p*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        dur[r['Kernel_Name'].split('(')[0]].append(float(r['End_Timestamp'])-float(r['Start_Timestamp']))
out={}
for k,v in acc.items():
    if not any(s in k for s in ('k_match4_b','k_accum_b','k_solve_b')): continue
    out[k]={c:{'launches':len(x),'mean':sum(x)/len(x)} for c,x in v.items()}
    out[k]['duration_us_under_pmc']=sum(dur[k])/max(1,len(dur[k]))/1e3
json.dump(out, open('gpurun_out/r04/pmc_valu.json','w'), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
