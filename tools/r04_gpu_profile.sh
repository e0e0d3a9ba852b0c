#!/bin/bash
# Round-4 measurement pass on the GPU box (run through gpurun from the repo root), in two calls to stay inside one call's limit:
#   bash tools/r04_gpu_profile.sh a1    # default bench, kernel traces, counters of the dominant kernels
#   bash tools/r04_gpu_profile.sh a2    # encoder GEMM counters, log-mel, cached-K/V comparison runs
#   bash tools/r04_gpu_profile.sh a3    # cached vs absorbed over output lengths and batch sizes (tools/r04_cross_sweep.sh)
#   bash tools/r04_gpu_profile.sh b     # fine-tune step, sizing runs
# Everything lands under gpurun_out/r04/; tools/r04_summaries.py turns it into the committed summaries under profiles/.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-finetune"
if [ "$1" = "a3" ]; then
echo "== cached vs absorbed over output lengths and batch sizes"
bash $ROOT/tools/r04_cross_sweep.sh || exit 1
echo "== done a3"; exit 0
fi
if [ "$1" = "a" ] || [ "$1" = "a1" ] || [ "$1" = "a2" ]; then
if [ "$1" != "a2" ]; then
echo "== default bench"; python3 $ROOT/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
echo "== kernel trace of the default bench command"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --steps 6 > $OUT/kt.log 2>&1 || exit 1
echo "== kernel trace, one pass in flight"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt1 -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --steps 3 --pipeline 1 > $OUT/kt1.log 2>&1 || exit 1
echo "== counters: fused cross block on cached K / V (cross_attention=cached)"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_c1 -- python3 $ROOT/tools/pmc_cross_block.py > $OUT/pmc_c1.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_c2 -- python3 $ROOT/tools/pmc_cross_block.py > $OUT/pmc_c2.log 2>&1 || exit 1
python3 $ROOT/tools/pmc_cross_block.py > $OUT/pmc_c_timing.log 2>&1 || exit 1
echo "== counters: streaming kernel of the absorbed cross-attention (the dominant kernel of the default decode step)"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_x1 -- python3 $ROOT/tools/pmc_cross_absorbed.py > $OUT/pmc_x1.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_x2 -- python3 $ROOT/tools/pmc_cross_absorbed.py > $OUT/pmc_x2.log 2>&1 || exit 1
python3 $ROOT/tools/pmc_cross_absorbed.py > $OUT/pmc_x_timing.log 2>&1 || exit 1
if [ "$1" = "a1" ]; then echo "== done a1"; cat $OUT/bench_default.json; exit 0; fi
fi
echo "== counters: encoder GEMMs"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_g1 -- python3 $ROOT/tools/pmc_gemm.py > $OUT/pmc_g1.log 2>&1 || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/pmc_g2 -- python3 $ROOT/tools/pmc_gemm.py > $OUT/pmc_g2.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_g3 -- python3 $ROOT/tools/pmc_gemm.py > $OUT/pmc_g3.log 2>&1 || echo "(LDS counter set not available)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_g -o gemm -- python3 $ROOT/tools/pmc_gemm.py > $OUT/kt_g.log 2>&1 || exit 1
echo "== log-mel: trace and HBM counters"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_lm -o lm -- python3 $ROOT/tools/logmel_bench.py 64 80 > $OUT/kt_lm.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_lm1 -- python3 $ROOT/tools/logmel_bench.py 64 80 > $OUT/pmc_lm1.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_lm2 -- python3 $ROOT/tools/logmel_bench.py 64 80 > $OUT/pmc_lm2.log 2>&1 || exit 1
python3 $ROOT/tools/logmel_bench.py 64 80 > $OUT/logmel_80.log 2>&1 || exit 1
python3 $ROOT/tools/logmel_bench.py 64 128 > $OUT/logmel_128.log 2>&1 || exit 1
echo "== cached K / V cross-attention (opt-in) for comparison: 64 and 224 new tokens, with the default beside it"
$B --cross-attention cached > $OUT/bench_cached.json 2> /dev/null || exit 1
$B --cross-attention cached --new-tokens 224 --steps 6 > $OUT/bench_cached_n224.json 2> /dev/null || exit 1
$B --new-tokens 224 --steps 6 > $OUT/size_small_n224.json 2> /dev/null || exit 1   # cross_attention=auto: picks cached K / V here (64 clips, >= 192 new tokens)
$B --cross-attention absorbed --new-tokens 224 --steps 6 > $OUT/size_small_n224_absorbed.json 2> /dev/null || exit 1
if [ "$1" = "a" ]; then
echo "== cached vs absorbed over output lengths and batch sizes"
bash $ROOT/tools/r04_cross_sweep.sh || exit 1
fi
echo "== done $1"
else
echo "== fine-tune step"
python3 $ROOT/bench.py --mode train --steps 5 --warmup 1 > $OUT/train_exact.json 2> $OUT/train_exact.err || exit 1
python3 $ROOT/bench.py --mode train --steps 5 --warmup 1 --f32 split > $OUT/train_split.json 2> $OUT/train_split.err || exit 1
echo "== medium B=256";          $B --model medium --batch 256 --pipeline 2 --steps 4 > $OUT/size_medium_b256.json 2> /dev/null || exit 1
echo "== medium B=256 cached";   $B --model medium --batch 256 --pipeline 2 --steps 4 --cross-attention cached > $OUT/size_medium_b256_cached.json 2> /dev/null || exit 1
echo "== large-v3 B=128 bf16";   $B --model large-v3 --batch 128 --pipeline 2 --steps 4 > $OUT/size_large_b128_bf16.json 2> /dev/null || exit 1
echo "== large-v3 B=128 fp8 w";  $B --model large-v3 --batch 128 --pipeline 2 --steps 4 --weights fp8 > $OUT/size_large_b128_fp8.json 2> /dev/null || exit 1
echo "== large-v3 B=128 fp8 w+a"; $B --model large-v3 --batch 128 --pipeline 2 --steps 4 --weights fp8 --activations fp8 > $OUT/size_large_b128_fp8_act.json 2> /dev/null || exit 1
echo "== small fp8";             $B --weights fp8 > $OUT/size_small_fp8.json 2> /dev/null || exit 1
echo "== small pipeline 1";      $B --pipeline 1 --steps 4 > $OUT/size_small_p1.json 2> /dev/null || exit 1
for f in $OUT/size_*.json $OUT/train_*.json; do echo "$(basename $f): $(python3 -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['value'], d.get('decode_step',{}).get('ms_per_step'), d.get('roofline',{}).get('frac'), d.get('roofline_mfma',{}).get('frac'))")"; done
fi
