#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace --stats -d DIR -o NAME`): total and
average duration per (kernel, grid).  usage: prof_db_summary.py <results.db> [header text]"""
import collections
import re
import sqlite3
import sys

db = sys.argv[1]
if len(sys.argv) > 2:
    print("# " + sys.argv[2])
c = sqlite3.connect(db)
agg = collections.defaultdict(lambda: [0, 0.0])
for name, gx, gy, gz, wx, st, en in c.execute("select name, grid_x, grid_y, grid_z, workgroup_x, start, end from kernels"):
    short = name.replace("(anonymous namespace)::", "").replace("void ", "")
    short = re.sub(r"\(.*", "", short)
    short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", short)
    key = (short[:88], f"{gx // max(wx, 1)}x{gy}x{gz}")
    agg[key][0] += 1
    agg[key][1] += (en - st) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"# total kernel time {tot / 1e3:.2f} ms over {sum(v[0] for v in agg.values())} dispatches")
for (k, g), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k:90s} grid={g:>14s} calls={n:6d} total_ms={t / 1e3:9.2f} avg_us={t / n:9.2f}")
