#!/bin/bash
# launch-latency knobs and pipeline depth (GPU box): each line = env / flags, value, ms/pass, single-pass ms, decode-step ms
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  env "$@" python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --steps 8 $FLAGS 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$* $FLAGS ->', d['value'], d['ms_per_step'], d['ms_per_pass_single_in_flight'], d['decode_step']['ms_per_step'])"
}
FLAGS=""
run X=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run GPU_MAX_HW_QUEUES=4
run GPU_MAX_HW_QUEUES=16
for p in 3 5 6; do FLAGS="--pipeline $p"; run X=1; done
FLAGS="--pipeline 6"; run GPU_MAX_HW_QUEUES=16
FLAGS="--pipeline 8"; run GPU_MAX_HW_QUEUES=16
