#!/usr/bin/env python3
"""How much do kernels of different HIP streams overlap in a rocprofv3 kernel trace?"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = []
per_stream = collections.Counter()
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, 1)); ev.append((e, -1))
    per_stream[(r["Stream_Id"], r["Queue_Id"])] += 1
ev.sort()
busy = [0, 0, 0, 0, 0]
cur = 0; last = ev[0][0]
for t, d in ev:
    busy[min(cur, 4)] += t - last
    last = t; cur += d
tot = ev[-1][0] - ev[0][0]
print("streams/queues:", dict(per_stream))
print("span ms", tot / 1e6, " idle %.1f%%  1 kernel %.1f%%  2 kernels %.1f%%  3 %.1f%%  4+ %.1f%%" % tuple(100 * b / tot for b in busy))
