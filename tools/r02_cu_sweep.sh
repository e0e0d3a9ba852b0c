#!/bin/bash
# CU partition sweep: log-mel + encoder (MFMA-bound) on HIP streams limited to N CUs (hipExtStreamCreateWithCUMask), the decode
# loops on unrestricted streams.  WIPA_BENCH_ENC_STREAMS = number of CU-limited encoder streams the passes in flight rotate over
# (each is its own hardware queue: with the 4 pooled queues and the null stream, more than 3 of them oversubscribe the 8 queue
# slots and the passes serialise).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r02_cu_sweep.txt; : > $out
run() { echo "## WIPA_BENCH_ENC_STREAMS=$WIPA_BENCH_ENC_STREAMS $*" >> $out; python bench.py --no-cpu-baseline --no-finetune --steps 12 --warmup 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['ms_per_pass_single_in_flight'])" >> $out; }
export WIPA_BENCH_ENC_STREAMS=3
run
for n in 256 248 240 232 224 192 160 128; do run --encoder-cus $n; done
run --encoder-cus 192 --pipeline 3
cat $out
