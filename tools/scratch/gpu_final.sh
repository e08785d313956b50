set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04 gpurun_out/prof
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_final.json 2> gpurun_out/r04/bench_final.err
tail -8 gpurun_out/r04/bench_final.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04/bench_final.json').read().strip().splitlines()[-1])
print('BENCH', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_of_floor'), d['roofline']['avg_launch_ms'], d['cpu_baseline'])
ss = d.get('single_sequence') or {}
print('SS', ss.get('value'), ss.get('whole_run_scans_per_s'), ss.get('ratio_vs_cpu_driver_c_only'), ss.get('cpu_driver'))
print('SS stages', ss.get('ms_per_scan_by_stage'))
print('SS ate', ss.get('ate_rmse_origin_m'), ss.get('ate_origin_pct_of_path'), '1M', ss.get('with_1M_point_local_map'))
nd = d.get('single_sequence_ndt') or {}
print('NDT', nd.get('value'), nd.get('whole_run_scans_per_s'), nd.get('ratio_vs_cpu_driver_c_only'), (nd.get('cpu_driver') or {}).get('value_c_library_only'), nd.get('ate_origin_pct_of_path'))
print('MULTI', d.get('multi_sequence'))
print('CREAL', (d.get('creal') or {}).get('value'), (d.get('creal') or {}).get('cpu_baseline'))
PY
SKIP_ODOM= timeout 2400 bash profiles/collect.sh r04 2>&1 | tail -45
