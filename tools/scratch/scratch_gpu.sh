for S in 24 32 48 64; do
python bench.py --streams $S --steps 16 --warmup 3 --no-cpu-baseline --no-shared-run --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('S', d['config']['scans_per_step_per_gpu'], 'value %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'match us/scan %.2f' % (1e3*d['roofline']['avg_kernel_ms_per_scan']))"
done
