"""Stream-level timeline of ONE timed step from a rocprofv3 --kernel-trace CSV of bench.py: per HIP queue busy time, union busy
time, idle gaps, time with >= 2 kernels in flight, and (optionally) the kernel-by-kernel listing of a window.
usage: python tools/timeline.py <kernel_trace.csv> [list_from_ms list_to_ms]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
# step boundaries: the last adamw dispatch of each run of adamw dispatches
last = [e for j, e in enumerate(ends) if j + 1 == len(ends) or ends[j + 1] - e > 2]
lo, hi = last[-2] + 1, last[-1] + 1
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
T = (int(step[-1]["End_Timestamp"]) - t0) / 1e6
qkey = "Queue_Id" if "Queue_Id" in step[0] else "Stream_Id"
queues = {}
for r in step:
    queues.setdefault(r[qkey], []).append(r)
print(f"step span {T:.3f} ms, {len(step)} dispatches, {len(queues)} queues")
ev = []
for q, rs in sorted(queues.items(), key=lambda kv: -len(kv[1])):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e6
    a = (int(rs[0]["Start_Timestamp"]) - t0) / 1e6
    b = (int(rs[-1]["End_Timestamp"]) - t0) / 1e6
    names = {}
    for r in rs:
        n = r["Kernel_Name"].split("(")[0].split("::")[-1][:28]
        names[n] = names.get(n, 0) + 1
    top = ", ".join(f"{k} x{v}" for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:4])
    print(f"  queue {q}: {len(rs):4d} kernels, busy {busy:7.3f} ms, first start {a:7.3f}, last end {b:7.3f} | {top}")
    for r in rs:
        ev.append((int(r["Start_Timestamp"]), 1))
        ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
depth, prev, t_idle, t_one, t_multi = 0, t0, 0, 0, 0
for t, d in ev:
    dt = t - prev
    if depth == 0:
        t_idle += dt
    elif depth == 1:
        t_one += dt
    else:
        t_multi += dt
    depth += d
    prev = t
print(f"  no kernel in flight {t_idle / 1e6:.3f} ms | exactly one {t_one / 1e6:.3f} ms | two or more {t_multi / 1e6:.3f} ms")
if len(sys.argv) > 3:
    a, b = float(sys.argv[2]), float(sys.argv[3])
    qid = {q: i for i, q in enumerate(sorted(queues, key=lambda q: -len(queues[q])))}
    for r in step:
        s = (int(r["Start_Timestamp"]) - t0) / 1e6
        if a <= s <= b:
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
            print(f"    {s:8.3f} ms  q{qid[r[qkey]]}  {d:8.1f} us  {n}")
