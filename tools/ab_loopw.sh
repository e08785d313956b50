#!/bin/bash
# runs tools/multi_seq_bench.py for every tools/variants/libmolahip_<name>.so given (and the product library as "base"):  tools/ab_loopw.sh "1,8,16" name...
COUNTS=$1; shift
for v in base "$@"; do
  if [ $v = base ]; then unset LD_LIBRARY_PATH MOLAHIP_LIB_PATH; else
    mkdir -p /tmp/mhvar_$v && cp tools/variants/libmolahip_$v.so /tmp/mhvar_$v/libmolahip.so
    export LD_LIBRARY_PATH=/tmp/mhvar_$v MOLAHIP_LIB_PATH=/tmp/mhvar_$v/libmolahip.so
  fi
  echo "== $v (MH_LOOPW=${MH_LOOPW:-default})"
  timeout 600 python tools/multi_seq_bench.py 400 $COUNTS 2>&1 | grep "^threads" | python3 -c "
import sys,json
for l in sys.stdin:
    n=l.split()[1]; r=json.loads(l.split(' ',2)[2]); p=r['ms_per_scan_sequence_0']
    print('  seq %2s steady %7.0f run_icp %.3f onLidar %.3f identical %s' % (n, r['steady_scans_per_s'], p['onLidar.3.run_icp'], p['onLidar'], r.get('trajectories_identical_to_solo_run')))
"
done
