#!/bin/bash
# registers / scratch / LDS of the kernels of an object file's gfx950 code object:  tools/kernel_resources.sh [mh_icp.o] [name filter]
OBJ=${1:-/root/repo/mola_lidar_odometry_amd/csrc/mh_icp.o}
FIL=${2:-.}
T=$(mktemp -d); cp "$OBJ" $T/o.o; (cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading o.o >/dev/null 2>&1)
CO=$(ls $T | grep gfx950 | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/$CO | python3 -c "
import sys,re
cur={}
out=[]
for l in sys.stdin:
    m=re.match(r'\s+[-\s]*\.(\w+):\s+(.*)',l)
    if not m: continue
    k,v=m.group(1),m.group(2).strip()
    if k=='name' and 'kd' not in v and cur.get('name') is None: cur['name']=v
    if k in('vgpr_count','sgpr_count','agpr_count','vgpr_spill_count','sgpr_spill_count','private_segment_fixed_size','group_segment_fixed_size'): cur[k]=v
    if k=='symbol':
        cur['symbol']=v; out.append(cur); cur={}
import subprocess
for c in out:
    n=subprocess.run(['c++filt',c.get('symbol','').replace('.kd','')],capture_output=True,text=True).stdout.strip()[:60]
    print(f\"{n:60s} vgpr={c.get('vgpr_count')} agpr={c.get('agpr_count')} sgpr={c.get('sgpr_count')} spill={c.get('vgpr_spill_count')} scratch={c.get('private_segment_fixed_size')} lds={c.get('group_segment_fixed_size')}\")
" | grep -E "$FIL"
rm -rf $T
