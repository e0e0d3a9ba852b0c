#!/bin/bash
# session 8: encoder GEMMs on a persistent grid of G workgroups (= CUs), the rest of the chip left to the decode loops of the other passes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04s8
mkdir -p $OUT
WIPA_GEMM_MAX_WGS=192 timeout -k 10 400 python3 -m pytest $ROOT/tests/test_gpu_kernels.py -q -x -k "gemm" > $OUT/persist_test.log 2>&1 || { tail -20 $OUT/persist_test.log; exit 1; }
tail -1 $OUT/persist_test.log
for cfg in "0 2" "256 2" "224 2" "192 2" "160 2" "128 2" "192 1" "160 1" "192 4"; do set -- $cfg
  WIPA_GEMM_MAX_WGS=$1 timeout -k 10 200 python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --cross-splits $2 --steps 16 > $OUT/persist_$1_$2.json 2>$OUT/persist_$1_$2.err || { tail -5 $OUT/persist_$1_$2.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$OUT/persist_$1_$2.json').read().strip().splitlines()[-1]); print('gemm workgroups $1 splits $2:', d['ms_per_step'], d['value'], d['ms_per_pass_single_in_flight'], d['roofline_mfma']['gemm_ms_per_pass'], d['passes_identical'])"
done
for g in 0 256 192; do WIPA_GEMM_MAX_WGS=$g timeout -k 10 200 python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --phase enc --pipeline 1 --steps 16 2>/dev/null | tail -1; done
