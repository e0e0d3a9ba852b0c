#!/bin/bash
# session 8: streaming-kernel shapes (waves x slots per workgroup) x frame splits; tests first
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04s8
mkdir -p $OUT
for w in 3 1 23; do
  WIPA_ABS_WAVES=$w timeout -k 10 300 python3 -m pytest $ROOT/tests/test_gpu_kernels.py -q -x -k "absorbed" > $OUT/waves_test_$w.log 2>&1 || { echo "tests failed for waves=$w"; tail -15 $OUT/waves_test_$w.log; exit 1; }
  echo "waves $w: $(tail -1 $OUT/waves_test_$w.log)"
done
for cfg in "3 4" "3 2" "1 4" "1 2" "23 2" "23 4" "2 4" "2 2"; do set -- $cfg
  WIPA_ABS_WAVES=$1 timeout -k 10 200 python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --cross-splits $2 --steps 12 > $OUT/waves_$1_$2.json 2>$OUT/waves_$1_$2.err || { tail -5 $OUT/waves_$1_$2.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$OUT/waves_$1_$2.json').read().strip().splitlines()[-1]); r=d['roofline']; print('waves $1 splits $2:', d['ms_per_step'], d['value'], d['ms_per_pass_single_in_flight'], d['decode_step']['ms_per_step'], r['avg_launch_ms'], r.get('two_launches_side_by_side',{}).get('avg_pair_ms'), d['passes_identical'])"
done
