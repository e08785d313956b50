#!/bin/bash
# Collects the evidence the bench numbers rest on (run on the GPU box through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the default bench command      -> gpurun_out/prof/<tag>_bench_*
#   2. PMC passes of the SAME configuration as the timed region (S = 32 scans per lock-step launch, a map per scan), each
#      in its own run with --kernel-trace only: FETCH_SIZE, WRITE_SIZE, L2 hit/miss, SQ busy/wait/VALU, TCP requests
#   3. the odometry driver on a synthetic drive (kernel stats of the whole per-scan path)
# Every profiler run has its own timeout and queues the uploads from the main thread (--upload-thread 0: a second host thread
# under rocprofv3's dispatch interception stalled a whole collection run).
# Usage: profiles/collect.sh <tag> [extra bench args]     (then copy gpurun_out/prof/<tag>_* into profiles/)
TAG=${1:-r02}
shift
EXTRA="$@"
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o ${TAG}_bench -- python $REPO/bench.py --upload-thread 0 --no-extras $EXTRA > $OUT/${TAG}_bench_stdout.log 2>&1
if [ -z "$SKIP_ODOM" ]; then
# the odometry driver on the first 200 scans of the synthetic city drive (the drive of bench.py's single_sequence extra)
python -c "
import sys; sys.path.insert(0, '$REPO')
from mola_lidar_odometry_amd import synth_city
print(synth_city.write_kitti_drive('$OUT/city', 200, time_channel=True)[0])" > $OUT/${TAG}_city_dir.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/odom -o ${TAG}_odom -- $REPO/mola_lidar_odometry_amd/molahip-lo-cli --pipeline $REPO/pipelines/lidar3d-default-hip.yaml --seq-dir $(cat $OUT/${TAG}_city_dir.txt) --time-field 12 --profile --out $OUT/${TAG}_odom.tum > $OUT/${TAG}_odom_stdout.log 2>&1
# ... and the NDT pipeline on the same scans (config-5 proxy)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/odom_ndt -o ${TAG}_odom_ndt -- $REPO/mola_lidar_odometry_amd/molahip-lo-cli --pipeline $REPO/pipelines/lidar3d-ndt-hip.yaml --seq-dir $(cat $OUT/${TAG}_city_dir.txt) --time-field 12 --profile --out $OUT/${TAG}_odom_ndt.tum > $OUT/${TAG}_odom_ndt_stdout.log 2>&1
rm -rf $OUT/city
fi
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
         "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_THREAD_CYCLES_VALU" \
         "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
         "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TD_TD_BUSY_sum" \
         "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc$i -o ${TAG}_pmc$i -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-shared-run --no-extras --upload-thread 0 $EXTRA > $OUT/${TAG}_pmc$i.log 2>&1
done
cd $REPO
python - <<PY
import csv, glob, collections, json, os
out="$OUT"; tag="$TAG"
acc=collections.defaultdict(lambda: collections.defaultdict(list))
dur=collections.defaultdict(list)
for f in glob.glob(out+'/pmc*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob(out+'/pmc*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        dur[r['Kernel_Name'].split('(')[0]].append(float(r['End_Timestamp'])-float(r['Start_Timestamp']))
summary={}
for k,v in acc.items():
    if not any(s in k for s in ('k_match','k_accum','k_solve','k_cov','k_tile','k_compact','k_count')): continue
    summary[k]={c:{'launches':len(x),'mean':sum(x)/len(x)} for c,x in v.items()}
    if dur.get(k): summary[k]['duration_us_under_pmc']={'launches':len(dur[k]),'mean':sum(dur[k])/len(dur[k])/1e3}
res={'counters':summary}
# the kernel the bench times: the lock-step match launch (one launch over all scans of a step)
mk=[k for k in summary if k.startswith('k_match') and k.endswith('_b')] or [k for k in summary if 'k_match' in k]
stdout=open(out+'/'+tag+'_bench_stdout.log').read().strip().splitlines()
line=[l for l in stdout if l.startswith('{')]
bench=json.loads(line[-1]) if line else {}
S=bench.get('roofline',{}).get('scans_per_launch') or bench.get('config',{}).get('scans_per_step_per_gpu')  # scans in ONE lock-step launch (a step is several launches' worth)
if mk:
    k=max(mk, key=lambda n: summary[n].get('SQ_WAVES',{}).get('mean',0))
    m=summary[k]
    t={'kernel':k,'scans_per_launch':S}
    # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream
    # (MI355X_MICROARCH.md, HBM section) -> x2 on the read side
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
        t['hbm_bytes_per_launch']=(2.0*m['FETCH_SIZE']['mean']+m['WRITE_SIZE']['mean'])*1024.0
        t['hbm_note']='(2 x FETCH_SIZE + WRITE_SIZE) KiB per launch; FETCH_SIZE doubled per the gfx950 guide; Infinity-Cache hits are counted as traffic'
    if 'TCP_TCC_READ_REQ_sum' in m: t['l2_request_bytes_per_launch']=m['TCP_TCC_READ_REQ_sum']['mean']*64.0
    if 'SQ_INSTS_VALU' in m: t['valu_wave_instructions_per_launch']=m['SQ_INSTS_VALU']['mean']
    # round 5: the ceilings of the bench line's views (clock cycles of the launch from GRBM_GUI_ACTIVE over the 8 XCDs)
    if 'GRBM_GUI_ACTIVE' in m:
        cyc=m['GRBM_GUI_ACTIVE']['mean']/8.0
        t['launch_cycles_under_pmc']=cyc
        if 'TA_TA_BUSY_sum' in m: t['ta_busy_frac']=m['TA_TA_BUSY_sum']['mean']/256.0/cyc
        if 'TD_TD_BUSY_sum' in m: t['td_busy_frac']=m['TD_TD_BUSY_sum']['mean']/256.0/cyc
    for c in ('SQ_INSTS_VMEM_RD','SQ_INSTS_VMEM_WR','SQ_INSTS_LDS','SQ_INSTS_LDS_ATOMIC','SQ_INSTS_SALU','SQ_WAVES','TCP_TOTAL_CACHE_ACCESSES_sum',
              'TCP_PENDING_STALL_CYCLES_sum','TCP_GATE_EN1_sum','SQ_LDS_IDX_ACTIVE','SQ_THREAD_CYCLES_VALU'):
        if c in m: t[c]=m[c]['mean']
    t['valu_mix']={c[len('SQ_INSTS_VALU_'):]:m[c]['mean'] for c in m if c.startswith('SQ_INSTS_VALU_')}
    if 'SQ_WAIT_ANY' in m and 'SQ_WAVE_CYCLES' in m: t['wait_frac']=m['SQ_WAIT_ANY']['mean']/m['SQ_WAVE_CYCLES']['mean']
    if 'TCC_HIT_sum' in m and 'TCC_MISS_sum' in m: t['l2_hit_rate']=m['TCC_HIT_sum']['mean']/max(1.0,m['TCC_HIT_sum']['mean']+m['TCC_MISS_sum']['mean'])
    if 'TCP_TOTAL_CACHE_ACCESSES_sum' in m and 'TCP_TCC_READ_REQ_sum' in m: t['l1_hit_rate']=1.0-m['TCP_TCC_READ_REQ_sum']['mean']/max(1.0,m['TCP_TOTAL_CACHE_ACCESSES_sum']['mean'])
    # duration of that kernel in the --stats run of the default bench (no counters): what the roofline divides by
    rows=list(csv.DictReader(open(glob.glob(out+'/bench/'+tag+'_*kernel_stats.csv')[0])))
    for r in rows:
        if r['Name'].split('(')[0]==k: t['avg_launch_ms']=float(r['AverageNs'])/1e6; t['calls']=int(r['Calls'])
    res['timed_kernel']=t
json.dump(res, open(out+'/'+tag+'_pmc_summary.json','w'), indent=1)
print(json.dumps(res.get('timed_kernel'), indent=1))
rows=list(csv.DictReader(open(glob.glob(out+'/bench/'+tag+'_*kernel_stats.csv')[0])))
for r in rows[:10]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(9),'us', r['Percentage'])
print(stdout[-1][:3000] if stdout else 'no bench output')
if glob.glob(out+'/odom/'+tag+'_*kernel_stats.csv'):
    print('--- odometry driver, 200 scans of the synthetic city drive (~115k raw points each)')
    for r in list(csv.DictReader(open(glob.glob(out+'/odom/'+tag+'_*kernel_stats.csv')[0])))[:14]:
        print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(9),'us', r['Percentage'])
    print(open(out+'/'+tag+'_odom_stdout.log').read().strip()[-1200:])
PY
# gpurun merges at most 64 MiB back: the raw per-dispatch tables stay on the box, the summaries travel
rm -rf $OUT/pmc[0-9]* 
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*agent_info.csv' -delete
du -sh $OUT
