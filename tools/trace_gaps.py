"""development aid: idle gaps between consecutive lock-step kernels of a bench run (rocprofv3 --kernel-trace CSV)."""
import csv
import glob
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "")))
rows.sort()
batch = [r for r in rows if r[2] in ("k_match4_b", "k_accum_b", "k_solve_b")]
gaps = []
for a, b in zip(batch, batch[1:]):
    g = (b[0] - a[1]) / 1e3
    if g < 300:  # (step boundaries are longer)
        gaps.append((g, a[2], b[2]))
bound = [(b[0] - a[1]) / 1e3 for a, b in zip(batch, batch[1:]) if (b[0] - a[1]) / 1e3 >= 300]
print("step-boundary gaps (us):", [round(g) for g in bound])
# what runs inside the boundary gaps
for a, b in zip(batch, batch[1:]):
    if (b[0] - a[1]) / 1e3 >= 300:
        inside = [(r[2][:28] or "copy/none", round((r[0] - a[1]) / 1e3), round((r[1] - r[0]) / 1e3)) for r in rows if a[1] <= r[0] < b[0]]
        print("  boundary:", inside[:14])
        break
tot = sum(g for g, _, _ in gaps)
n_match = sum(1 for r in batch if r[2] == "k_match4_b")
dur = {k: sum((r[1] - r[0]) for r in batch if r[2] == k) / 1e3 / max(1, sum(1 for r in batch if r[2] == k)) for k in ("k_match4_b", "k_accum_b", "k_solve_b")}
print("launches", len(batch), "match launches", n_match, "avg us", {k: round(v, 1) for k, v in dur.items()})
print("sum of in-step gaps: %.1f us per match launch (%.0f us per 20-iteration step); gaps > 10 us: %d" % (tot / max(1, n_match), 20 * tot / max(1, n_match), sum(1 for g, _, _ in gaps if g > 10)))
big = sorted(gaps, reverse=True)[:8]
print("largest:", [(round(g, 1), a, b) for g, a, b in big])
other = [r for r in rows if r[2] not in ("k_match4_b", "k_accum_b", "k_solve_b")]
names = {}
for r in other:
    names[r[2]] = names.get(r[2], 0) + 1
print("other kernels:", sorted(names.items(), key=lambda kv: -kv[1])[:8])
