#!/bin/bash
# Round-2 sizing runs and the cross-block counters (run through gpurun from the repo root): bash tools/r02_sizing.sh
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== cross block counters"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_x1 -- python3 $ROOT/tools/pmc_cross_block.py > $OUT/pmc_x1.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_x2 -- python3 $ROOT/tools/pmc_cross_block.py > $OUT/pmc_x2.log 2>&1 || exit 1
python3 $ROOT/tools/pmc_cross_block.py > $OUT/pmc_x_timing.log 2>&1 || exit 1
B="python3 $ROOT/bench.py --no-cpu-baseline --no-finetune"
echo "== 224 new tokens";        $B --new-tokens 224 --steps 6 > $OUT/size_small_n224.json 2> /dev/null || exit 1
echo "== medium B=256";          $B --model medium --batch 256 --pipeline 2 --steps 4 > $OUT/size_medium_b256.json 2> /dev/null || exit 1
echo "== large-v3 B=128 bf16";   $B --model large-v3 --batch 128 --pipeline 2 --steps 4 > $OUT/size_large_b128_bf16.json 2> /dev/null || exit 1
echo "== large-v3 B=128 fp8";    $B --model large-v3 --batch 128 --pipeline 2 --steps 4 --weights fp8 > $OUT/size_large_b128_fp8.json 2> /dev/null || exit 1
echo "== small fp8";             $B --weights fp8 > $OUT/size_small_fp8.json 2> /dev/null || exit 1
echo "== small pipeline 1";      $B --pipeline 1 --steps 4 > $OUT/size_small_p1.json 2> /dev/null || exit 1
for f in $OUT/size_*.json; do echo "$(basename $f): $(python3 -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['value'], d.get('decode_step',{}).get('ms_per_step'), d.get('decode_step',{}).get('frac'), d.get('roofline',{}).get('frac'), d.get('roofline_mfma',{}).get('frac'))")"; done
