"""scratch: the few numbers of a bench.py output line that a probe run is after."""
import json, sys
for f in sys.argv[1:]:
    lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
    if not lines:
        print(f, "NO JSON:", open(f).read()[-600:])
        continue
    d = json.loads(lines[-1])
    r = d.get("roofline") or {}
    print(f, "value %.0f scans/s  ms/step %.3f  S=%s  io=%s  match us/scan %.2f  alone us %.2f  frac %.3f  parity %s" % (
        d["value"], d["ms_per_step"], d["config"]["scans_per_step_per_gpu"], d["config"]["timed_region"][:28],
        1e3 * (r.get("avg_kernel_ms_per_scan") or 0), 1e3 * (r.get("avg_kernel_ms_alone") or 0), r.get("frac") or 0,
        d.get("parity_vs_cpu")))
