#!/usr/bin/env python
"""Summarise rocprofv3 `--pmc ... --output-format csv` runs: mean counter value per (kernel, grid, counter).
usage: pmc_summary.py out.csv dir1/p_counter_collection.csv [dir2/...]   (kernels filtered by SDV_PMC_FILTER, default all)"""
import collections
import csv
import os
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)


flt = os.environ.get("SDV_PMC_FILTER", "")
agg = collections.OrderedDict()
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if flt and flt not in k:
            continue
        key = (k, r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["LDS_Block_Size"], r["Counter_Name"])
        agg.setdefault(key, []).append(float(r["Counter_Value"]))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402  (csrc_fingerprint: which kernel sources these counters were collected from - bench.py refuses to replay others)

with open(sys.argv[1], "w") as f:
    f.write(f"# csrc={bench.csrc_fingerprint()} rocprofv3 --pmc passes of tools/unet_once.py, summarised by tools/pmc_summary.py\n")
    w = csv.writer(f)
    w.writerow(["kernel", "grid", "workgroup", "vgpr", "lds_bytes", "counter", "dispatches", "mean", "min", "max"])
    for key, v in agg.items():
        w.writerow(list(key) + [len(v), f"{sum(v) / len(v):.6g}", f"{min(v):.6g}", f"{max(v):.6g}"])
print(f"{len(agg)} rows -> {sys.argv[1]}")
