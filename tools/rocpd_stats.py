#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2 rocpd SQLite) kernel trace into the classic --stats CSV:
name, calls, total ns, average ns, percentage.  usage: rocpd_stats.py results.db [out.csv]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*\)$", "", name)


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                      "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
    for n, c, t, a, mn, mx in rows:
        lines.append(f"\"{short(n)}\",{c},{t},{a:.0f},{mn},{mx},{100.0 * t / total:.2f}")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    sys.stdout.write(out)


if __name__ == "__main__":
    main()
