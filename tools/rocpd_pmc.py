#!/usr/bin/env python
"""Per-kernel sums of the PMC counters in a rocprofv3 rocpd database.  usage: rocpd_pmc.py results.db [out.csv]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", name))


db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
print("# columns:", cols, file=sys.stderr)
rows = db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name "
                  "order by kernel_name").fetchall() if "kernel_name" in cols else []
out = ["Kernel,Counter,Dispatches,Sum"]
for k, c, n, v in rows:
    out.append(f"\"{short(k)}\",{c},{n},{v}")
txt = "\n".join(out) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt)
sys.stdout.write(txt)
