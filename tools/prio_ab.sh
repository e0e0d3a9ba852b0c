#!/bin/bash
# A/B on one box: the pipelined pass with every pass's encoder and decode loop on ONE default-priority stream (shipped) against
# encoder on a default-priority stream + decode loop on a HIGH-priority stream (WIPA_PIPE_PRIO=1).  bash tools/prio_ab.sh
B="python3 bench.py --no-cpu-baseline --no-finetune --no-other-configs --steps 16"
for rep in 1 2; do for P in 0 2; do
  printf "WIPA_PIPE_PRIO=%s : " $P
  WIPA_PIPE_PRIO=$P timeout -k 10 300 $B 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'],'ms per pass', d['value'],'audio-s/s | identical', d['passes_identical'], '| single', d['ms_per_pass_single_in_flight'])" || exit 1
done; done
for P in 0 2; do printf "dec phase, 64 clips, WIPA_PIPE_PRIO=%s : " $P; WIPA_PIPE_PRIO=$P timeout -k 10 200 python3 bench.py --phase dec --batch 64 --pipeline 4 --steps 8 --warmup 1 2>/dev/null | tail -1 | cut -c1-80; done
