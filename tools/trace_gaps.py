#!/usr/bin/env python
"""Timeline summary of one infer() step from a rocprofv3 --kernel-trace csv: per kernel class busy time, and the idle gaps between
consecutive kernels (launch boundaries).  usage: trace_gaps.py <kernel_trace.csv> [n_last_steps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one step = from one preprocess_kernel to the next
idx = [i for i, r in enumerate(rows) if "preprocess_kernel" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
step = rows[a:b]
t0, t1 = int(step[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
gaps = [int(step[i + 1]["Start_Timestamp"]) - int(step[i]["End_Timestamp"]) for i in range(len(step) - 1)]
print(f"step wall {(t1 - t0) / 1e3:.1f} us, kernels {len(step)}, busy {busy / 1e3:.1f} us, gaps total {sum(gaps) / 1e3:.1f} us (mean {sum(gaps) / len(gaps) / 1e3:.2f} us, max {max(gaps) / 1e3:.1f})")
cls = collections.defaultdict(lambda: [0, 0])
for r in step:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0][:60]
    cls[n][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cls[n][1] += 1
for n, (t, c) in sorted(cls.items(), key=lambda kv: -kv[1][0]):
    print(f"  {n:60s} {t / 1e3:9.1f} us  x{c:3d}  avg {t / c / 1e3:7.1f}")
