for v in "" "MH_LOCKSTEP_GROUPS=2" "MH_LOCKSTEP_GROUPS=2" ""; do
  env $v python bench.py --no-extras --no-cpu-baseline --no-shared-run 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],4))"
done
