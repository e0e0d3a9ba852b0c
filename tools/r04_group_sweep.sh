#!/bin/bash
# session 8: --decode-group G (the decode loops of G consecutive 64-clip passes as one 64 G-row loop; encoder still per pass) -- NOT the reported configuration
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04s8
mkdir -p $OUT
for cfg in "2 4 2" "2 3 2" "2 2 2" "2 4 1" "4 2 1" "4 3 1" "4 4 1" "4 2 2"; do set -- $cfg
  timeout -k 10 300 python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --decode-group $1 --pipeline $2 --cross-splits $3 --steps 24 > $OUT/group_$1_$2_$3.json 2>$OUT/group_$1_$2_$3.err || { tail -5 $OUT/group_$1_$2_$3.err; continue; }
  python3 -c "
import json; d=json.loads(open('$OUT/group_$1_$2_$3.json').read().strip().splitlines()[-1]); print('group $1 groups in flight $2 splits $3:', d['ms_per_step'], d['value'])"
done
