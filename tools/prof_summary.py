#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) totals.  usage: prof_summary.py <dir> [passes]"""
import collections, csv, glob, re, sys

d = sys.argv[1]
passes = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    m = re.search(r"(gemm_nt_kernel|gemm_skinny_kernel|decode_attn_kernel|attn_generic_kernel|layernorm_kernel|flash_enc_bf16_kernel|greedy_step_kernel|embed_kernel|mel_log_kernel|mel_norm_kernel|reflect_pad_kernel|advance_pos_kernel)", n)
    short = m.group(1) if m else n[:40]
    if "gemm" in short:
        short += "<" + ("bf16" if "DF16b" in n.split("Params")[0][-30:] or "_Accum" in n else "f32") + ">"
    key = (short, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Workgroup_Size_X"])
    agg[key][0] += 1
    agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"total kernel ms per pass: {tot / 1e3 / passes:.2f}")
for (k, g, w), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{k:32s} blocks={g:>7d} wg={w:>5s} calls/pass={c / passes:8.1f} ms/pass={t / 1e3 / passes:8.2f} avg_us={t / c:9.2f}")
