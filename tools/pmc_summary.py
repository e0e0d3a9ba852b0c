#!/usr/bin/env python3
"""Pivot a rocprofv3 --pmc counter_collection CSV: mean counter value per (kernel, grid).  usage: pmc_summary.py <dir> [filter]"""
import collections, csv, glob, sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt and flt not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"][:60], r.get("Grid_Size", r.get("Grid_Size_X", "")))
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, ctr in rows.items():
    print(key[0], "grid", key[1])
    for name, vals in sorted(ctr.items()):
        print(f"   {name:36s} n={len(vals):3d} mean={sum(vals) / len(vals):.4g}")
