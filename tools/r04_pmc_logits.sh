#!/bin/bash
# session 8: the logits projection plain (wipa_gemm) and lean (wipa_logits_greedy, no logit stores): event timing and HBM counters
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04s8
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for g in 0 2 1; do GREEDY=$g python3 $ROOT/tools/pmc_logits.py | tail -2 | head -1; done
for g in 0 1; do
GREEDY=$g rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_l${g}f -- python3 $ROOT/tools/pmc_logits.py > $OUT/pmc_l${g}f.log 2>&1 || exit 1
GREEDY=$g rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_l${g}w -- python3 $ROOT/tools/pmc_logits.py > $OUT/pmc_l${g}w.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
for g in ("0", "1"):
    for c, d in (("FETCH_SIZE", "f"), ("WRITE_SIZE", "w")):
        v = []
        for f in glob.glob("$OUT/pmc_l%s%s/**/*counter_collection.csv" % (g, d), recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm_wide_persistent" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    v.append(float(r["Counter_Value"]))
        print("GREEDY=%s %s: %d launches, mean %.1f KB" % (g, c, len(v), sum(v) / max(len(v), 1)))
PY
