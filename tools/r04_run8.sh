#!/bin/bash
# round 4, session 8: the frame-split setting of the absorbed streaming launch.  tests of the new setting, the default bench (2 splits
# with 4 passes in flight), the cached-vs-absorbed table again with 2 splits, and the streaming kernel's HBM counters at 2 splits.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04s8
mkdir -p $OUT
python3 -m pytest $ROOT/tests/test_gpu_kernels.py $ROOT/tests/test_gpu_model.py $ROOT/tests/test_gpu_bench.py -m gpu -x -q -k "splits or absorbed or bench" > $OUT/tests_new.log 2>&1 || { tail -30 $OUT/tests_new.log; exit 1; }
tail -2 $OUT/tests_new.log
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-finetune"
show() { python3 - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], d["config"]["cross_attention"], d["config"].get("cross_frame_splits"), d["ms_per_step"], d["value"],
      d.get("ms_per_pass_single_in_flight"), d.get("decode_step", {}).get("ms_per_step"), d.get("default_splits"), d["passes_identical"])
PY
}
$B > $OUT/bench_s2.json 2> $OUT/bench_s2.err || exit 1; show $OUT/bench_s2.json
$B --cross-splits 4 > $OUT/bench_s4.json 2> /dev/null || exit 1; show $OUT/bench_s4.json
$B --cross-splits 3 > $OUT/bench_s3.json 2> /dev/null || exit 1; show $OUT/bench_s3.json
for n in 32 128 224; do
  st=12; [ $n = 224 ] && st=6
  $B --new-tokens $n --steps $st --cross-attention absorbed > $OUT/n${n}_abs_s2.json 2> /dev/null || exit 1; show $OUT/n${n}_abs_s2.json
  $B --new-tokens $n --steps $st --cross-attention cached > $OUT/n${n}_cached.json 2> /dev/null || exit 1; show $OUT/n${n}_cached.json
done
$B --batch 128 --cross-attention absorbed --steps 6 > $OUT/b128_abs_s2.json 2> /dev/null || exit 1; show $OUT/b128_abs_s2.json
$B --batch 128 --cross-attention absorbed --cross-splits 4 --steps 6 > $OUT/b128_abs_s4.json 2> /dev/null || exit 1; show $OUT/b128_abs_s4.json
echo "== counters: streaming kernel at 2 splits"
SPLITS=2 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_x1 -- python3 $ROOT/tools/pmc_cross_absorbed.py > $OUT/pmc_x1.log 2>&1 || exit 1
SPLITS=2 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_x2 -- python3 $ROOT/tools/pmc_cross_absorbed.py > $OUT/pmc_x2.log 2>&1 || exit 1
SPLITS=2 python3 $ROOT/tools/pmc_cross_absorbed.py > $OUT/pmc_x_timing.log 2>&1 || exit 1
tail -2 $OUT/pmc_x_timing.log
echo "== done"
