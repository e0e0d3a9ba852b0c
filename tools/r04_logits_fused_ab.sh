#!/bin/bash
# session 8: logits projection with greedy partials (default) against the plain GEMM + row-scanning tail (WIPA_LOGITS_FUSED=0), same box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for m in 1 0 1 0; do WIPA_LOGITS_FUSED=$m python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --steps 24 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WIPA_LOGITS_FUSED=$m:', d['ms_per_step'], d['value'], d['ms_per_pass_single_in_flight'], d['decode_step']['ms_per_step'], d['default_splits']['ms_per_pass_single_in_flight'], d['default_splits']['decode_step_ms'], d['tokens_checksum'], d['passes_identical'])"; done
