set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04 gpurun_out/prof
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_final.json 2> gpurun_out/r04/bench_final.err; tail -8 gpurun_out/r04/bench_final.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04/bench_final.json'))
print('BENCH', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_of_floor'), d['roofline']['avg_launch_ms'], d['cpu_baseline'])
s=d['single_sequence']; print('SS', s['value'], s['whole_run_scans_per_s'], s['ratio_vs_cpu_driver'], s['cpu_driver'])
print('SS stages', s['ms_per_scan_by_stage'])
n=d['single_sequence_ndt']; print('NDT', n['value'], n['whole_run_scans_per_s'], n['ratio_vs_cpu_driver'], n['cpu_driver'].get('value_c_library_only'))
print('MULTI', d['multi_sequence']); print('CREAL', d['creal']['value'], d['creal']['cpu_baseline'])
PY
SKIP_ODOM= timeout 2400 bash profiles/collect.sh r04 2>&1 | tail -45
