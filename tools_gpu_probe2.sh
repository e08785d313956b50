#!/bin/bash
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 600 python bench.py 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value',round(d['value'],1),'ms/step',round(d['ms_per_step'],2)); print('roofline',{k:(round(v,4) if isinstance(v,float) else v) for k,v in d['roofline'].items() if k!='note'}); print('cpu',d['cpu_baseline']); print('parity',d['parity_vs_cpu'])"
