#!/usr/bin/env python
"""rocprofv3 --kernel-trace CSV (p_kernel_trace.csv) -> the classic --stats table: name, calls, total / average / min / max ns, %.
usage: trace_stats.py p_kernel_trace.csv out.csv"""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)


agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(short(r["Kernel_Name"]), [0, 0, 1 << 62, 0])
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
with open(sys.argv[2], "w") as f:
    f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage\n")
    for n, (c, t, mn, mx) in rows:
        f.write(f"\"{n}\",{c},{t},{t / c:.0f},{mn},{mx},{100.0 * t / tot:.2f}\n")
for n, (c, t, mn, mx) in rows[:12]:
    print(f"{100.0 * t / tot:6.2f} %  {c:7d} x {t / c / 1e3:9.1f} us  {n}")
