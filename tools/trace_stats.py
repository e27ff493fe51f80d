"""Per-kernel statistics of the TIMED steps only, from a rocprofv3 --kernel-trace CSV of bench.py.
Step boundaries are recognised by the optimizer: every EgoClip step ends with a run of adamw_kernel dispatches; the last K
steps are kept (warm-up and the instrumented extra passes before them are dropped).
usage: python tools/trace_stats.py <kernel_trace.csv> <K timed steps> <out.csv>"""
import collections
import csv
import sys

path, K, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = []                       # index of the last adamw dispatch of every step
for i, r in enumerate(rows):
    if "adamw_kernel" in r["Kernel_Name"]:
        if ends and i - ends[-1] <= 2:
            ends[-1] = i
        else:
            ends.append(i)
if len(ends) < K + 1:
    raise SystemExit(f"only {len(ends)} optimizer steps in the trace, need {K + 1}")
lo, hi = ends[-K - 1] + 1, ends[-1] + 1
acc = collections.OrderedDict()
for r in rows[lo:hi]:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    acc.setdefault(r["Kernel_Name"], []).append(d)
tot = sum(sum(v) for v in acc.values())
span = int(rows[hi - 1]["End_Timestamp"]) - int(rows[lo]["Start_Timestamp"])
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow([f"# {K} timed steps: {hi - lo} dispatches, kernel time {tot / K / 1e6:.3f} ms per step, "
                f"span {span / K / 1e6:.3f} ms per step (rocprofv3 --kernel-trace of bench.py, warm-up dropped)"])
    w.writerow(["Name", "CallsPerStep", "TotalNsPerStep", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([name, round(len(v) / K, 2), round(sum(v) / K), round(sum(v) / len(v), 1), round(100.0 * sum(v) / tot, 3), min(v), max(v)])
print(open(out).readline().strip())
