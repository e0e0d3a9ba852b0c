#!/usr/bin/env python3
"""Where the time of a decode step goes when several passes are in flight, from rocprofv3 kernel traces.

usage: python tools/decode_gaps.py <trace_pipeline4.csv> <trace_pipeline1.csv>      (rocprofv3 --kernel-trace --output-format csv
       of `bench.py --phase dec --batch 8 --pipeline P`; tools/profile.sh gaps collects both)

For every HIP stream: the dispatches in start order; gap = start of kernel N+1 - end of kernel N (same stream), split by whether a
kernel of ANOTHER stream started inside the gap.  Kernels of a decode step form a dependent chain, so on one stream
step time = sum of durations + sum of gaps.  Printed: per predecessor kernel (what the gap waits behind) the mean duration and
mean gap with 1 and with P passes in flight, and the totals per decode step."""
import csv
import sys
from collections import defaultdict


def kernel_name(raw: str) -> str:
    """a readable, stable label: the function's own name plus its integer template arguments.  rocprofv3 writes some names mangled
    (_ZN12_GLOBAL__N_1<len><name>I...E...) and some demangled (void (anonymous namespace)::name<...>(...))."""
    import re

    m = re.match(r"_ZN?(?:12_GLOBAL__N_1)?(\d+)", raw)
    if m:
        n, at = int(m.group(1)), m.end()
        name, rest = raw[at:at + n], raw[at + n:]
        ints = re.findall(r"L[ib](\d+)E", rest.split("EEv")[0]) if rest.startswith("I") else []
        return name + ("<" + ",".join(ints) + ">" if ints else "")
    raw = raw.replace("void ", "").replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in raw:  # cut at the argument list: the first "(" outside the template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out)


def load(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            name = kernel_name(r["Kernel_Name"])
            rows.append((int(r["Stream_Id"]), int(r["Queue_Id"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    return rows


def analyse(rows):
    """the decode streams = those carrying the step graph's kernels; the timed region = the last 60 % of each stream's dispatches
    (warm-up, graph capture and the first passes are in the head)"""
    by_stream = defaultdict(list)
    for s, q, t0, t1, n in rows:
        by_stream[s].append((t0, t1, n, q))
    dec = {s: sorted(v) for s, v in by_stream.items() if sum("greedy_tail" in x[2] for x in v) > 50}
    starts_all = sorted((t0, s) for s, v in dec.items() for t0, _, _, _ in v)
    import bisect
    st_times = [t for t, _ in starts_all]
    per_kernel = defaultdict(lambda: [0, 0.0, 0.0, 0, 0.0])  # n, dur, gap, gaps with a foreign start inside, gap time of those
    steps = 0
    span = 0.0
    queues = defaultdict(set)
    for s, v in dec.items():
        v = v[int(len(v) * 0.4):]
        for i in range(len(v) - 1):
            t0, t1, n, q = v[i]
            queues[s].add(q)
            gap = v[i + 1][0] - t1
            if gap > 200_000 or gap < -50_000:  # pass boundary (host collects / relaunches) or overlapping graph branches: not chain gaps
                continue
            lo, hi = bisect.bisect_right(st_times, t1), bisect.bisect_left(st_times, v[i + 1][0])
            foreign = any(starts_all[j][1] != s for j in range(lo, hi))
            k = per_kernel[n]
            k[0] += 1
            k[1] += t1 - t0
            k[2] += max(gap, 0)
            if foreign:
                k[3] += 1
                k[4] += max(gap, 0)
            if "greedy_tail" in n:
                steps += 1
        span += v[-1][1] - v[0][0]
    return per_kernel, steps, len(dec), span, queues


def main():
    p4, p1 = sys.argv[1], sys.argv[2]
    A, stepsA, nsA, spanA, qA = analyse(load(p4))
    B, stepsB, nsB, spanB, qB = analyse(load(p1))
    print(f"decode streams: {nsA} (trace 1) / {nsB} (trace 2); decode steps counted: {stepsA} / {stepsB}")
    print(f"hardware queues seen per decode stream: {sorted(len(v) for v in qA.values())} / {sorted(len(v) for v in qB.values())}")
    print(f"wall per step on a stream IN THESE TRACES: {spanA / stepsA / 1e3:.1f} us with {nsA} in flight, {spanB / stepsB / 1e3:.1f} us alone "
          f"(the tracer runs the passes one after another: {nsA} x the lone figure, not the untraced behaviour)")
    print(f"\n{'kernel (the gap is AFTER it)':58s} {'per step':>8s} | {'dur 1':>7s} {'dur P':>7s} | {'gap 1':>7s} {'gap P':>7s} | {'gaps w/ foreign start':>21s} {'their mean':>10s}")
    tot = defaultdict(float)
    for n in sorted(A, key=lambda n: -(A[n][1] + A[n][2])):
        a, b = A[n], B.get(n)
        if b is None or a[0] < 20:
            continue
        per_step = a[0] / stepsA
        d1, dP, g1, gP = b[1] / b[0] / 1e3, a[1] / a[0] / 1e3, b[2] / b[0] / 1e3, a[2] / a[0] / 1e3
        fr = a[3] / a[0]
        print(f"{n[:58]:58s} {per_step:8.1f} | {d1:7.2f} {dP:7.2f} | {g1:7.2f} {gP:7.2f} | {fr:21.2f} {(a[4] / a[3] / 1e3 if a[3] else 0):10.2f}")
        tot["d1"] += d1 * per_step
        tot["dP"] += dP * per_step
        tot["g1"] += g1 * per_step
        tot["gP"] += gP * per_step
    print(f"\nper decode step (us): kernel time {tot['d1']:.0f} alone -> {tot['dP']:.0f} with P in flight; gaps {tot['g1']:.0f} -> {tot['gP']:.0f}")


if __name__ == "__main__":
    main()
