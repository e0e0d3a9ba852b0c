python -m pytest tests -m gpu -x -q > gpurun_out/r4_t3.log 2>&1; tail -3 gpurun_out/r4_t3.log
for v in 1 2 0; do WIPA_MERGE_SINGLE=$v python -m pytest tests/test_gpu_kernels.py -q -k "four_streams" > gpurun_out/r4_merge_single_$v.log 2>&1; echo "WIPA_MERGE_SINGLE=$v: $(tail -1 gpurun_out/r4_merge_single_$v.log)"; done
python bench.py --no-finetune --no-cpu-baseline > gpurun_out/r4_b3.json 2> gpurun_out/r4_b3.err
WIPA_ABS_MERGE_OUT=0 python bench.py --no-finetune --no-cpu-baseline > gpurun_out/r4_b3_nomergeout.json 2> gpurun_out/r4_b3_nomergeout.err
WIPA_ABS_PROLOGUE_CLIPS=8 python bench.py --no-finetune --no-cpu-baseline > gpurun_out/r4_b3_p8.json 2> gpurun_out/r4_b3_p8.err
python - <<PY
import json
for f in ("r4_b3","r4_b3_nomergeout","r4_b3_p8"):
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["ms_per_pass_single_in_flight"], d["decode_step"]["ms_per_step"], d["passes_identical"], d["roofline"]["avg_launch_ms"])
PY
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_p1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-finetune --steps 3 --pipeline 1 > $GRAFT_REPO_ROOT/gpurun_out/r4_prof_p1.json 2> $GRAFT_REPO_ROOT/gpurun_out/r4_prof_p1.err; ls $GRAFT_REPO_ROOT/gpurun_out/prof_p1 | head
