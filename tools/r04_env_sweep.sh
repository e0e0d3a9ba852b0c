#!/bin/bash
# session 8: runtime environment knobs that change how launches / graph nodes reach the GPU (A/B on one box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() { echo -n "$1: "; env $1 timeout -k 10 200 python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --steps 24 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['ms_per_pass_single_in_flight'], d['default_splits']['decode_step_ms'], d['passes_identical'])" || echo failed; }
run "X=0"
run "HIP_FORCE_DEV_KERNARG=1"
run "HIP_FORCE_DEV_KERNARG=0"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run "HSA_ENABLE_INTERRUPT=0"
run "ROC_SIGNAL_POOL_SIZE=1024"
run "X=0"
