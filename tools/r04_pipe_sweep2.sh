#!/bin/bash
# passes in flight x hardware queues at 2 frame splits (session 8)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04s8
mkdir -p $OUT
for cfg in "4 4" "3 4" "5 4" "5 8" "6 8" "8 8" "4 8"; do set -- $cfg
  GPU_MAX_HW_QUEUES=$2 timeout -k 10 200 python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --pipeline $1 --steps 24 > $OUT/pipe_$1_q$2.json 2>/dev/null || exit 1
  python3 -c "
import json; d=json.loads(open('$OUT/pipe_$1_q$2.json').read().strip().splitlines()[-1]); print('passes $1 queues $2:', d['ms_per_step'], d['value'], d['config']['cross_frame_splits'], d['passes_identical'])"
done
