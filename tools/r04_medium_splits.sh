ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for s in 4 2; do
timeout -k 10 500 python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --model medium --batch 256 --pipeline 2 --steps 3 --cross-splits $s > $ROOT/gpurun_out/r04s8_medium_s$s.json 2>/dev/null || exit 1
python3 -c "
import json; d=json.loads(open('$ROOT/gpurun_out/r04s8_medium_s$s.json').read().strip().splitlines()[-1]); print('medium 256 clips, 2 in flight, splits $s:', d['ms_per_step'], d['value'], d['ms_per_pass_single_in_flight'], d['decode_step']['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
