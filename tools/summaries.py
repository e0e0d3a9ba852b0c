#!/usr/bin/env python3
"""Turn gpurun_out/$ROUND/ (tools/profile.sh bench | gemm | cross | train-size) into the committed summaries under profiles/.
usage: ROUND=r05 SPLITS=2 python tools/summaries.py     (SPLITS: the frame splits the streaming kernel's counters were taken at ->
profiles/<round>_pmc_cross_absorbed.json for 4, ..._s<SPLITS>.json otherwise; CROSS_TABLE=1 also rewrites the cached-vs-absorbed table)"""
import collections
import csv
import glob
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("ROUND", "r05")
SRC = os.path.join(ROOT, "gpurun_out", os.environ.get("SRC", ROUND))
SPLITS = int(os.environ.get("SPLITS", "2"))
DST = os.environ.get("DST") or os.path.join(ROOT, "profiles")  # on the GPU box: gpurun_out/<round>/ (tools/profile.sh)
os.makedirs(DST, exist_ok=True)


def counters(d):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(SRC, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            key = (r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "")))
            rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in ctr.items()} for k, ctr in rows.items()}


def trace_us(d, pattern):
    dur = []
    for f in glob.glob(os.path.join(SRC, d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if pattern in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return dur


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:80]


def have(*parts):
    return os.path.exists(os.path.join(SRC, *parts))


# ---- kernel stats of the default bench command (4 passes in flight) and of --pipeline 1
for tag, d, cmd in (("default_cmd", "kt", "python3 bench.py --no-cpu-baseline --no-finetune --no-other-configs --steps 6"),
                    ("pipeline1", "kt1", "python3 bench.py --no-cpu-baseline --no-finetune --no-other-configs --steps 3 --pipeline 1")):
    if not have(d, "bench_kernel_stats.csv"):
        continue
    shutil.copy(os.path.join(SRC, d, "bench_kernel_stats.csv"), os.path.join(DST, f"{ROUND}_bench_{tag}_kernel_stats.csv"))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(os.path.join(SRC, d, "bench_kernel_trace.csv"))):
        key = (short(r["Kernel_Name"]), f'{int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]}')
        agg[key][0] += 1
        agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    with open(os.path.join(DST, f"{ROUND}_bench_{tag}_by_grid.txt"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- {cmd}   (MI355X, {ROUND}; all passes incl. warm-up, the evaluate-style run and the roofline microbenches; the tracer runs the passes in flight one after another)\n")
        f.write(f"# total kernel time {sum(v[1] for v in agg.values()) / 1e3:.2f} ms over {sum(v[0] for v in agg.values())} dispatches\n")
        for (k, g), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:44]:
            f.write(f"{k:82s} grid={g:>14s} calls={n:6d} total_ms={t / 1e3:9.2f} avg_us={t / n:9.2f}\n")

# ---- the dominant kernel: streaming kernel of the absorbed cross-attention
if have("pmc_x1"):
    x = {}
    n_launch = []
    for d in ("pmc_x1", "pmc_x2"):
        for k, c in counters(d).items():
            if "cross_absorbed_v2" in k[0]:
                for cn, (v, n) in c.items():
                    x[cn] = v
                    n_launch.append(n)
    log = open(os.path.join(SRC, "pmc_x_timing.log")).read()
    alg = int(re.search(r"algorithmic bytes per launch.*?:\s*(\d+)", log).group(1))
    ev = float(re.search(r"cross_absorbed_v2_kernel: ([0-9.]+) us", log).group(1))
    d1, d2 = trace_us("pmc_x1", "cross_absorbed_v2"), trace_us("pmc_x2", "cross_absorbed_v2")
    out = {"kernel": "cross_absorbed_v2_kernel<768> (decode-step cross-attention of one layer on the encoder output itself: scores with the key "
                     "projection absorbed into the query, P x xa with the value projection applied after the merge)",
           "shape": "whisper-small, B=64, H=12, Tk=1500, bf16, " + str(SPLITS) + " frame splits: 48 launches: 12 consecutive (the layers of a step) per encoder output, 4 encoder outputs in turn",
           "command": ("" if SPLITS == 4 else f"SPLITS={SPLITS} ") + "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 tools/pmc_cross_absorbed.py ; same with --pmc WRITE_SIZE (separate passes)",
           "FETCH_SIZE_KB_per_launch": round(x.get("FETCH_SIZE", 0), 2), "WRITE_SIZE_KB_per_launch": round(x.get("WRITE_SIZE", 0), 2),
           "launches_counted": n_launch,
           "correction": "gfx950: FETCH_SIZE counts a wide coalesced 16 B/lane stream at exactly 1/2 of its bytes (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE is exact",
           "hbm_bytes_per_launch": int(x.get("FETCH_SIZE", 0) * 1024 * 2 + x.get("WRITE_SIZE", 0) * 1024),
           "algorithmic_bytes_per_launch": alg,
           "avg_us_under_counters": [round(sum(d1) / max(len(d1), 1), 2), round(sum(d2) / max(len(d2), 1), 2)],
           "avg_us_event_timed_no_counters": ev}
    out["ratio_traffic_over_algorithmic"] = round(out["hbm_bytes_per_launch"] / alg, 4)
    json.dump(out, open(os.path.join(DST, f"{ROUND}_pmc_cross_absorbed.json" if SPLITS == 4 else f"{ROUND}_pmc_cross_absorbed_s{SPLITS}.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))

# ---- the dominant kernel of the default step: fused cross block on the cached K / V
if have("pmc_c1"):
    x, n_launch = {}, []
    for d in ("pmc_c1", "pmc_c2"):
        for k, c in counters(d).items():
            if "decode_cross_block" in k[0]:
                for cn, (v, n) in c.items():
                    x[cn] = v
                    n_launch.append(n)
    log = open(os.path.join(SRC, "pmc_c_timing.log")).read()
    alg = int(re.search(r"algorithmic bytes per launch.*?:\s*(\d+)", log).group(1))
    ev = float(re.search(r"wipa_decode_cross_block: ([0-9.]+) us", log).group(1))
    d1, d2 = trace_us("pmc_c1", "decode_cross_block"), trace_us("pmc_c2", "decode_cross_block")
    out = {"kernel": "decode_cross_block_pre_kernel<bf16, 4> (slab sum + residual + cross_attn_ln + cross query + streaming cross-attention; the first "
                     "32 key and 32 value rows of every wave staged into LDS by LDS-DMA under the prologue), non-temporal K/V loads",
           "shape": "whisper-small, B=64, H=12, Tk=1500, bf16, 2 slabs: 24 launches cycling 12 layer caches and 12 query matrices",
           "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 tools/pmc_cross_block.py ; same with --pmc WRITE_SIZE (separate passes)",
           "FETCH_SIZE_KB_per_launch": round(x.get("FETCH_SIZE", 0), 2), "WRITE_SIZE_KB_per_launch": round(x.get("WRITE_SIZE", 0), 2),
           "launches_counted": n_launch,
           "correction": "gfx950: FETCH_SIZE counts a wide coalesced 16 B/lane stream at exactly 1/2 of its bytes (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE is exact",
           "hbm_bytes_per_launch": int(x.get("FETCH_SIZE", 0) * 1024 * 2 + x.get("WRITE_SIZE", 0) * 1024),
           "algorithmic_bytes_per_launch": alg,
           "note": "the doubling correction over-counts the part of the fetches that is not a wide coalesced stream (the 98 KB of query weights each of the 768 workgroups reads from L2 reach the fabric counter only on an L2 miss)",
           "avg_us_under_counters": [round(sum(d1) / max(len(d1), 1), 2), round(sum(d2) / max(len(d2), 1), 2)],
           "avg_us_event_timed_no_counters": ev}
    out["ratio_traffic_over_algorithmic"] = round(out["hbm_bytes_per_launch"] / alg, 4)
    json.dump(out, open(os.path.join(DST, f"{ROUND}_pmc_cross_block.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))

# ---- encoder GEMM counters + per-shape times
if have("pmc_g1"):
    def by_launch(d):
        """counters of the gemm_nt launches in dispatch order: [{counter: value}] (three launches per shape, five shapes)"""
        rows = collections.defaultdict(dict)
        names_ = {}
        for f in glob.glob(os.path.join(SRC, d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm_nt" in r["Kernel_Name"]:
                    rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = rows[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                    names_[int(r["Dispatch_Id"])] = (r["Kernel_Name"], r["Grid_Size"])
        order = sorted(rows)
        return [rows[i] for i in order], [names_[i] for i in order]

    sets = [by_launch(d) for d in ("pmc_g1", "pmc_g2", "pmc_g3") if have(d)]
    names = [("qk", "N=1536 K=768  bias+scale", 1536, 768), ("mlp1", "N=3072 K=768  bias+GELU", 3072, 768),
             ("out", "N=768  K=768  bias+f32 residual", 768, 768), ("mlp2", "N=768  K=3072 bias+f32 residual", 768, 3072),
             ("mlp2*", "N=768  K=3072 bf16 out, no residual", 768, 3072)]
    # durations per launch in launch order (3 launches per shape), from the run without counters
    dur = []
    for f in glob.glob(os.path.join(SRC, "kt_g", "**", "*kernel_trace.csv"), recursive=True):
        rows = sorted((r for r in csv.DictReader(open(f)) if "gemm_nt" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    gemm = {}
    with open(os.path.join(DST, f"{ROUND}_pmc_encoder_gemm.txt"), "w") as f:
        f.write("# rocprofv3 --pmc on tools/pmc_gemm.py (M = 96000 rows, the encoder GEMM shapes of whisper-small at B = 64; three launches per\n"
                "# shape, counters averaged over the 2nd and 3rd), MI355X, " + ROUND + ".\n"
                "# passes: {SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES},\n"
                "# {GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16}, {SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD}; durations from a\n"
                "# separate --kernel-trace run (no counters).  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs).\n")
        for i, (tag, label, N, K) in enumerate(names):
            c = {}
            kname = None
            for launches, knames in sets:
                if len(launches) < 3 * i + 3:
                    continue
                kname = knames[3 * i + 1]
                for cn in launches[3 * i + 1]:
                    c[cn] = (launches[3 * i + 1][cn] + launches[3 * i + 2].get(cn, launches[3 * i + 1][cn])) / 2
            if not c:
                continue
            util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (c["GRBM_GUI_ACTIVE"] / 8) if "GRBM_GUI_ACTIVE" in c else float("nan")
            parked = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
            us = dur[3 * i + 1: 3 * i + 3] if len(dur) >= 3 * i + 3 else []
            avg = sum(us) / len(us) if us else float("nan")
            tf = 2 * 96000 * N * K / avg / 1e6 if us else float("nan")
            gemm[tag] = {"mfma_busy": round(util, 3), "waves_parked": round(parked, 3), "us": round(avg, 1), "TF/s": round(tf, 1)}
            f.write(f"{short(kname[0])} grid {kname[1]}   [{tag} {label}]   {avg:.1f} us = {tf:.0f} TF/s\n")
            for cn, v in sorted(c.items()):
                f.write(f"   {cn:34s} {v:.4g}\n")
            f.write(f"   -> MFMA utilisation {util:.3f}, waves parked {parked:.3f}\n")
    json.dump({"source": f"profiles/{ROUND}_pmc_encoder_gemm.txt", "definition": "SQ_VALU_MFMA_BUSY_CYCLES per SIMD / GRBM_GUI_ACTIVE per XCD", "by_gemm": gemm},
              open(os.path.join(DST, f"{ROUND}_pmc_encoder_gemm.json"), "w"), indent=1)
    print(json.dumps(gemm, indent=1))

# ---- log-mel
if have("kt_lm"):
    lm = {}
    for r in csv.DictReader(open(glob.glob(os.path.join(SRC, "kt_lm", "**", "*kernel_stats.csv"), recursive=True)[0])):
        if any(s in r["Name"] for s in ("logmel", "mel_", "reflect", "gemm_nt", "fillBuffer")):  # fused kernel, clamp pass, memsets
            lm[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2)}
    fetch = sum(v for k, c in counters("pmc_lm1").items() for cn, (v, n) in c.items() if cn == "FETCH_SIZE" and ("logmel" in k[0] or "mel_norm" in k[0] or "mel_clamp" in k[0]))
    write = sum(v for k, c in counters("pmc_lm2").items() for cn, (v, n) in c.items() if cn == "WRITE_SIZE" and ("logmel" in k[0] or "mel_norm" in k[0] or "mel_clamp" in k[0]))
    per_kernel = collections.defaultdict(dict)
    for d in ("pmc_lm1", "pmc_lm2"):
        for k, c in counters(d).items():
            if "logmel" in k[0] or "mel_norm" in k[0] or "mel_clamp" in k[0]:
                per_kernel[short(k[0])].update({cn: round(v, 1) for cn, (v, n) in c.items()})
    out = {"what": "log-mel front-end for 64 clips x 30 s, 80 mels, bf16 output in the conv1 halo layout (tools/logmel_bench.py 64 80)",
           "kernels": lm, "event_timed": open(os.path.join(SRC, "logmel_80.log")).read().strip().splitlines()[-1],
           "event_timed_128_mels": open(os.path.join(SRC, "logmel_128.log")).read().strip().splitlines()[-1],
           "counters_KB_per_launch": per_kernel,
           "hbm_bytes_per_batch": {"fetch_uncorrected": int(fetch * 1024), "fetch_doubled": int(fetch * 2048), "write": int(write * 1024)},
           "algorithmic_bytes_per_batch": 64 * 2880000,
           "correction": "FETCH_SIZE doubled for wide coalesced 16 B/lane streams (MI355X_MICROARCH.md); the fused kernel reads the audio with 4-byte "
                         "coalesced loads, so the doubled figure is an upper bound"}
    json.dump(out, open(os.path.join(DST, f"{ROUND}_logmel.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))

# ---- cached K / V vs absorbed projections: one table
rows = []
cases = [("whisper-small, 64 clips, 32 new tokens", "sw_{m}_s64_n32.json"), ("whisper-small, 64 clips, 64 new tokens (headline)", None),
         ("whisper-small, 64 clips, 128 new tokens", "sw_{m}_s64_n128.json"), ("whisper-small, 64 clips, 224 new tokens", None),
         ("whisper-small, 128 clips, 64 new tokens", "sw_{m}_s128_n64.json"), ("whisper-small, 128 clips, 224 new tokens", "sw_{m}_s128_n224.json"),
         ("whisper-medium, 256 clips, 64 new tokens, 2 passes in flight", None), ("whisper-medium, 256 clips, 224 new tokens, 2 passes in flight", "sw_{m}_m256_n224.json")]
special = {1: ("bench_cached.json", "bench_default.json"), 3: ("bench_cached_n224.json", "size_small_n224_absorbed.json"),
           6: ("size_medium_b256_cached.json", "size_medium_b256.json")}
for i, (label, pat) in enumerate(cases):
    files = special[i] if pat is None else (pat.format(m="cached"), pat.format(m="absorbed"))
    if all(have(f) and os.path.getsize(os.path.join(SRC, f)) > 0 for f in files):
        c, a = (json.loads(open(os.path.join(SRC, f)).read().strip().splitlines()[-1]) for f in files)
        rows.append((label, c["ms_per_step"], c["value"], a["ms_per_step"], a["value"]))
# (CROSS_TABLE=1 rewrites the table from the sw_*.json runs of `tools/profile.sh cross`)
if rows and os.environ.get("CROSS_TABLE") == "1":
    with open(os.path.join(DST, f"{ROUND}_cached_vs_absorbed.txt"), "w") as f:
        f.write("# bench.py --no-cpu-baseline --no-finetune --cross-attention {cached,absorbed} [...]   (MI355X; tools/profile.sh crosseep.sh)\n")
        f.write(f"# {'workload':66s} {'cached ms':>10s} {'audio-s/s':>10s} {'absorbed ms':>12s} {'audio-s/s':>10s} {'absorbed vs cached':>19s}\n")
        for label, cm, cv, am, av in rows:
            f.write(f"  {label:66s} {cm:10.2f} {cv:10.0f} {am:12.2f} {av:10.0f} {100 * (av / cv - 1):+18.1f}%\n")
    print(open(os.path.join(DST, f"{ROUND}_cached_vs_absorbed.txt")).read())

for n in ("bench_default.json", "bench_cached.json", "bench_cached_n224.json", "size_medium_b256_cached.json", "train_exact.json", "train_split.json", "size_small_n224.json", "size_small_n224_absorbed.json", "size_medium_b256.json",
          "size_large_b128_bf16.json", "size_large_b128_fp8.json", "size_large_b128_fp8_act.json", "size_small_fp8.json", "size_small_p1.json"):
    if have(n) and os.path.getsize(os.path.join(SRC, n)) > 0:
        shutil.copy(os.path.join(SRC, n), os.path.join(DST, ROUND + "_" + n))
