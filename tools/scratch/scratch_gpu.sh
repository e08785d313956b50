python tools/multi_seq_bench.py 100 1,2,4,8,16 2>&1 | tail -1 > gpurun_out/r03_multi_sequence_one_process.json
python tools/multi_seq_bench.py 100 1,4,8 pipelines/lidar3d-ndt-hip.yaml 2>&1 | tail -1 > gpurun_out/r03_multi_sequence_one_process_ndt.json
python -c "
import json
for f in ('gpurun_out/r03_multi_sequence_one_process.json','gpurun_out/r03_multi_sequence_one_process_ndt.json'):
    d=json.load(open(f))['multi_sequence_one_process']
    for m in ('threads','fibers'):
        print(d['pipeline'], m, {k:(round(v.get('steady_scans_per_s',0)), v.get('trajectories_identical_to_solo_run')) for k,v in d[m].items()})
"
python tools/map_insert_time.py | tail -1
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
