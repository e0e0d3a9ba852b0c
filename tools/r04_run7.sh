#!/bin/bash
# round 4, GPU session 7: passes in flight / hardware queues re-swept on the round-4 kernel mix (r02 swept them on the cached-K/V mix)
O=gpurun_out/r4s7; mkdir -p $O
B="python bench.py --no-finetune --no-cpu-baseline"
for p in 3 4 5 6 8; do $B --pipeline $p --steps 12 > $O/p$p.json 2> /dev/null; done
for q in 4 16; do GPU_MAX_HW_QUEUES=$q $B --pipeline 4 --steps 12 > $O/q$q.json 2> /dev/null; done
GPU_MAX_HW_QUEUES=16 $B --pipeline 6 --steps 12 > $O/q16p6.json 2> /dev/null
GPU_MAX_HW_QUEUES=16 $B --pipeline 8 --steps 12 > $O/q16p8.json 2> /dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["ms_per_step"], d["value"], d["config"]["passes_in_flight"], d["config"]["hw_queues"], d["passes_identical"])
    except Exception as e:
        print(f, "FAILED", e)
PY
