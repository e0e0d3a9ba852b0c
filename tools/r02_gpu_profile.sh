#!/bin/bash
# Round-2 measurement pass on the GPU box (run through gpurun from the repo root):
#   bash tools/r02_gpu_profile.sh
# Everything lands under gpurun_out/r02/; the summaries that are judged are copied to profiles/ afterwards.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== default bench"; python3 $ROOT/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
echo "== kernel trace of the default bench command"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --steps 6 > $OUT/kt.log 2>&1 || exit 1
echo "== kernel trace, one pass in flight"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt1 -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --steps 3 --pipeline 1 > $OUT/kt1.log 2>&1 || exit 1
echo "== pmc encoder gemm"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_g1 -- python3 $ROOT/tools/pmc_gemm.py > $OUT/pmc_g1.log 2>&1 || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/pmc_g2 -- python3 $ROOT/tools/pmc_gemm.py > $OUT/pmc_g2.log 2>&1 || exit 1
echo "== pmc logits gemm"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_l1 -- python3 $ROOT/tools/pmc_logits.py > $OUT/pmc_l1.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_l2 -- python3 $ROOT/tools/pmc_logits.py > $OUT/pmc_l2.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_l3 -- python3 $ROOT/tools/pmc_logits.py > $OUT/pmc_l3.log 2>&1 || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_l4 -- python3 $ROOT/tools/pmc_logits.py > $OUT/pmc_l4.log 2>&1 || exit 1
echo "== fine-tune step"
python3 $ROOT/bench.py --mode train --steps 5 --warmup 1 > $OUT/train_exact.json 2> $OUT/train_exact.err || exit 1
python3 $ROOT/bench.py --mode train --steps 5 --warmup 1 --f32 split > $OUT/train_split.json 2> $OUT/train_split.err || exit 1
echo "== done"
cat $OUT/bench_default.json $OUT/train_exact.json $OUT/train_split.json
