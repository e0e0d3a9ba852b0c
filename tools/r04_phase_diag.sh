#!/bin/bash
# session 8 diagnostic: the two halves of the pass alone, 1 / 2 / 4 passes in flight, 2 and 4 frame splits
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04s8
mkdir -p $OUT
for cfg in "enc 1 2" "enc 2 2" "enc 4 2" "dec 1 4" "dec 1 2" "dec 2 4" "dec 2 2" "dec 3 2" "dec 4 4" "dec 4 2" "dec 4 1" "dec 4 3"; do set -- $cfg
  timeout -k 10 200 python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --phase $1 --pipeline $2 --cross-splits $3 --steps 16 2>/dev/null | tail -1
done
