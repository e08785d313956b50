bash profiles/collect.sh r03 > gpurun_out/r03_collect.log 2>&1
python tools/multi_seq_bench.py 100 1,2,4,8,16 > gpurun_out/r03_ms.log 2>&1; tail -1 gpurun_out/r03_ms.log > gpurun_out/r03_multi_sequence_one_process.json
python tools/multi_seq_bench.py 100 1,4,8 pipelines/lidar3d-ndt-hip.yaml > gpurun_out/r03_ms_ndt.log 2>&1; tail -1 gpurun_out/r03_ms_ndt.log > gpurun_out/r03_multi_sequence_one_process_ndt.json
python tools/multi_seq_trace.py 60 8 gpurun_out/r03_multi_sequence_trace.json > /dev/null 2>&1
python tools/map_insert_time.py > gpurun_out/r03_map_insert_time.log 2>&1
python bench.py > gpurun_out/r03_bench_final.json 2> gpurun_out/r03_bench_final.err
tail -c 300 gpurun_out/r03_bench_final.err
