#!/bin/bash
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO
python tools/multi_seq_bench.py 60 1 > /dev/null 2>&1   # writes the tree
D=/tmp/molahip_multi/sequences/00
for n in 1 8; do
args=""
for i in $(seq $n); do args="$args --seq-dir $D"; done
./mola_lidar_odometry_amd/molahip-lo-cli --pipeline pipelines/lidar3d-default-hip.yaml --out /tmp/o.tum --profile $args 2>&1 | grep -E "profile_ms|sequences" | head -3 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if 'profile_ms_per_scan' in d:
        print({k.replace('onLidar.',''):round(v,3) for k,v in d['profile_ms_per_scan'].items() if v>0.004 and not k.startswith('icp.') and not k.startswith('prefetch')})
    else: print(d)
"
done
