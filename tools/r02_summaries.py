#!/usr/bin/env python3
"""Turn gpurun_out/r02/ (tools/r02_gpu_profile.sh) into the committed summaries under profiles/.  usage: tools/r02_summaries.py"""
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r02")
DST = os.path.join(ROOT, "profiles")


def counters(d):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(SRC, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            key = (r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "")))
            rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in ctr.items()} for k, ctr in rows.items()}


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:80]


# ---- kernel stats of the default bench command (4 passes in flight) and of --pipeline 1
for tag, d, cmd in (("default_cmd", "kt", "python3 bench.py --no-cpu-baseline --no-finetune --steps 6"),
                    ("pipeline1", "kt1", "python3 bench.py --no-cpu-baseline --no-finetune --steps 3 --pipeline 1")):
    shutil.copy(os.path.join(SRC, d, "bench_kernel_stats.csv"), os.path.join(DST, f"r02_bench_{tag}_kernel_stats.csv"))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(os.path.join(SRC, d, "bench_kernel_trace.csv"))):
        key = (short(r["Kernel_Name"]), f'{int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]}')
        agg[key][0] += 1
        agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    with open(os.path.join(DST, f"r02_bench_{tag}_by_grid.txt"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- {cmd}   (MI355X, round 2; all passes incl. warm-up and the roofline microbenches)\n")
        f.write(f"# total kernel time {sum(v[1] for v in agg.values()) / 1e3:.2f} ms over {sum(v[0] for v in agg.values())} dispatches\n")
        for (k, g), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
            f.write(f"{k:82s} grid={g:>14s} calls={n:6d} total_ms={t / 1e3:9.2f} avg_us={t / n:9.2f}\n")

# ---- encoder GEMM counters
g1, g2 = counters("pmc_g1"), counters("pmc_g2")
names = ["qk      N=1536 K=768  bias+scale", "mlp1    N=3072 K=768  bias+GELU", "out     N=768  K=768  bias+f32 residual",
         "mlp2    N=768  K=3072 bias+f32 residual", "mlp2*   N=768  K=3072 bf16 out, no residual"]
gemm = {}
with open(os.path.join(DST, "r02_pmc_encoder_gemm.txt"), "w") as f:
    f.write("# rocprofv3 --pmc on tools/pmc_gemm.py (M = 96000 rows, the encoder GEMM shapes of whisper-small at B = 64), MI355X, round 2.\n"
            "# two passes: {SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES} and\n"
            "# {GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16}.  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs).\n")
    keys = [k for k in g1 if "gemm_nt" in k[0]]
    for i, k in enumerate(keys):
        c = dict(g1[k])
        c.update(g2.get(k, {}))
        util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (c["GRBM_GUI_ACTIVE"] / 8) if "GRBM_GUI_ACTIVE" in c else float("nan")
        parked = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        label = names[i] if i < len(names) else ""
        gemm[label.split()[0] if label else str(i)] = {"mfma_busy": round(util, 3), "waves_parked": round(parked, 3)}
        f.write(f"{short(k[0])} grid {k[1]}   [{label}]\n")
        for cn, v in sorted(c.items()):
            f.write(f"   {cn:34s} {v:.4g}\n")
        f.write(f"   -> MFMA utilisation {util:.3f}, waves parked {parked:.3f}\n")
json.dump({"source": "profiles/r02_pmc_encoder_gemm.txt", "definition": "SQ_VALU_MFMA_BUSY_CYCLES per SIMD / GRBM_GUI_ACTIVE per XCD", "by_gemm": gemm},
          open(os.path.join(DST, "r02_pmc_encoder_gemm.json"), "w"), indent=1)

# ---- logits GEMM counters (the launch the round-1 counter run never produced)
l = {}
for d in ("pmc_l1", "pmc_l2", "pmc_l3", "pmc_l4"):
    for k, c in counters(d).items():
        if "gemm_skinny" in k[0]:
            l.update(c)
M, N, K = 64, 51865, 768
alg = N * K * 2 + M * K * 2 + M * 51872 * 4
dur = []
for f in glob.glob(os.path.join(SRC, "pmc_l1", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_skinny" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {"kernel": "gemm_skinny_kernel<bf16,f32,4,4,4> as the decode-step logits projection (M=64, N=51865, K=768)",
       "command": "rocprofv3 --pmc <set> --output-format csv -- python3 tools/pmc_logits.py ; separate passes: FETCH_SIZE | WRITE_SIZE | "
                  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES | GRBM_GUI_ACTIVE",
       "counters_mean_per_launch": {k: round(v, 1) for k, v in sorted(l.items())},
       "FETCH_SIZE_KB_per_launch": l.get("FETCH_SIZE"), "WRITE_SIZE_KB_per_launch": l.get("WRITE_SIZE"),
       "correction": "gfx950: FETCH_SIZE counts a wide coalesced 16 B/lane stream at 1/2 of its bytes (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE exact",
       "hbm_bytes_per_launch": int(l.get("FETCH_SIZE", 0) * 1024 * 2 + l.get("WRITE_SIZE", 0) * 1024),
       "algorithmic_bytes_per_launch": alg,
       "avg_us_under_counters": round(sum(dur) / max(len(dur), 1), 2)}
out["ratio_traffic_over_algorithmic"] = round(out["hbm_bytes_per_launch"] / alg, 4)
json.dump(out, open(os.path.join(DST, "r02_pmc_logits_gemm.json"), "w"), indent=1)
for n in ("bench_default.json", "train_exact.json", "train_split.json"):
    shutil.copy(os.path.join(SRC, n), os.path.join(DST, "r02_" + n))
print(open(os.path.join(DST, "r02_pmc_encoder_gemm.txt")).read()[-1500:])
print(json.dumps(out, indent=1))
