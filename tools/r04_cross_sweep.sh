#!/bin/bash
# cached K / V against absorbed-projection cross-attention over output lengths and batch sizes (run through gpurun):
#   bash tools/r04_cross_sweep.sh     -> gpurun_out/r04/sw_<mode>_<case>.json; tools/r04_summaries.py tabulates them
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
B="python3 $ROOT/bench.py --no-cpu-baseline --no-finetune"
for mode in cached absorbed; do
  timeout -k 10 300 $B --cross-attention $mode --new-tokens 32 --steps 12 > $OUT/sw_${mode}_s64_n32.json 2>/dev/null || exit 1
  timeout -k 10 300 $B --cross-attention $mode --new-tokens 128 --steps 8 > $OUT/sw_${mode}_s64_n128.json 2>/dev/null || exit 1
  timeout -k 10 400 $B --cross-attention $mode --batch 128 --steps 6 > $OUT/sw_${mode}_s128_n64.json 2>/dev/null || exit 1
  timeout -k 10 400 $B --cross-attention $mode --batch 128 --new-tokens 224 --steps 4 > $OUT/sw_${mode}_s128_n224.json 2>/dev/null || exit 1
  timeout -k 10 500 $B --cross-attention $mode --model medium --batch 256 --pipeline 2 --new-tokens 224 --steps 3 > $OUT/sw_${mode}_m256_n224.json 2>/dev/null || exit 1
done
