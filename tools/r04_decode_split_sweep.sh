ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04s8
mkdir -p $OUT
for cfg in "1 2 4" "2 4 4" "2 2 4" "2 4 2" "2 2 3" "1 2 3 16"; do set -- $cfg
  GPU_MAX_HW_QUEUES=${4:-8} timeout -k 10 200 python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --decode-split $1 --cross-splits $2 --pipeline $3 --steps 24 > $OUT/ds_$1_$2_$3.json 2>$OUT/ds_$1_$2_$3.err || { tail -5 $OUT/ds_$1_$2_$3.err; continue; }
  python3 -c "
import json; d=json.loads(open('$OUT/ds_$1_$2_$3.json').read().strip().splitlines()[-1]); print('decode-split $1 cross-splits $2 passes $3:', d['ms_per_step'], d['value'], d['passes_identical'])"
done
