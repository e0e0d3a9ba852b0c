#!/bin/bash
# Measurement passes on the GPU box, one mode per gpurun call (each stays inside one call's limit):
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/profile.sh <mode>'        ROUND=r05 (default) names the output
#
#   bench        default bench line; kernel traces of the default command and of one pass in flight; HBM counters of the decode step's
#                dominant kernels (absorbed streaming kernel at SPLITS=2|4 frame splits, fused cross block on cached K / V)
#   gemm         encoder GEMM counters (MFMA busy, waits, LDS) + trace; log-mel trace + HBM counters; cached-K/V comparison runs
#   cross        cached vs absorbed cross-attention over output lengths and batch sizes
#   train-size   fine-tune step (exact / split f32 products) and the sizing runs (medium 256, large-v3 128 bf16 / fp8, small fp8 / p1)
#   gaps         decode phase with 4 and 1 passes in flight: untraced ms per pass, and kernel traces (tools/decode_gaps.py reads them)
#   chain-probe  tools/micro/chain_probe.hip: dependent-launch chains on 1-4 streams, timed from inside the kernels
#   gemm-probe   tools/micro/gemm_loop_probe.hip: the encoder GEMM's loop taken apart, + MFMA-busy counters of every variant
#   chain-pad    the chain probe and the real decode loops against GPU_MAX_HW_QUEUES x idle padding streams x the library's own streams
#   group-sweep  tools/group_sweep.py: decode groups x groups in flight
#   final        the tree's final check: pytest -m gpu, smoke(), the default bench line
# RAW rocprofv3 output stays on the GPU box (/tmp/wipa_prof/$ROUND: a traced bench run is > 64 MiB, more than gpurun copies back);
# every mode ends by running tools/summaries.py THERE, so what comes back under gpurun_out/$ROUND/ is the summaries (named as they
# are committed: copy them into profiles/) and the small JSON lines / logs.  profiles/README.md maps every committed profile to its mode.  Programs go directly after `--` under rocprofv3, and
# counter passes never share a run with a trace domain other than --kernel-trace.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=${ROUND:-r05}
SPLITS=${SPLITS:-2}
export SPLITS   # tools/pmc_cross_absorbed.py reads it: the frame splits of the streaming launch (2 = several passes in flight, 4 = lone decode)
KEEP=$ROOT/gpurun_out/$ROUND          # what gpurun copies back: summaries, JSON lines, logs
OUT=/tmp/wipa_prof/$ROUND             # raw traces and counter CSVs: stay on the box
mkdir -p $OUT $KEEP
MODE=$1
reduce() {  # raw -> summaries (in $KEEP, under their committed names) + the small files
  SRC=$OUT DST=$KEEP ROUND=$ROUND SPLITS=$SPLITS CROSS_TABLE=${CROSS_TABLE:-0} python3 $ROOT/tools/summaries.py > $KEEP/summaries_$MODE.log 2>&1 || { tail -5 $KEEP/summaries_$MODE.log; return 1; }
  find $OUT -maxdepth 1 -type f -size -2M \( -name "*.json" -o -name "*.txt" -o -name "*.log" -o -name "*.err" \) -exec cp {} $KEEP/ \;
  du -sh $KEEP | cut -f1
}
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --no-other-configs"
case "$MODE" in
bench)
  echo "== default bench"; python3 $ROOT/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
  echo "== kernel trace of the default bench command (the tracer runs the passes in flight one after another: durations, not concurrency)"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --no-other-configs --steps 6 > $OUT/kt.log 2>&1 || exit 1
  echo "== kernel trace, one pass in flight"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt1 -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --no-other-configs --steps 3 --pipeline 1 > $OUT/kt1.log 2>&1 || exit 1
  echo "== counters: streaming kernel of the absorbed cross-attention at $SPLITS frame splits (the dominant kernel of the decode step)"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_x1 -- python3 $ROOT/tools/pmc_cross_absorbed.py > $OUT/pmc_x1.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_x2 -- python3 $ROOT/tools/pmc_cross_absorbed.py > $OUT/pmc_x2.log 2>&1 || exit 1
  python3 $ROOT/tools/pmc_cross_absorbed.py > $OUT/pmc_x_timing.log 2>&1 || exit 1
  echo "== counters: fused cross block on cached K / V (cross_attention=cached)"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_c1 -- python3 $ROOT/tools/pmc_cross_block.py > $OUT/pmc_c1.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_c2 -- python3 $ROOT/tools/pmc_cross_block.py > $OUT/pmc_c2.log 2>&1 || exit 1
  python3 $ROOT/tools/pmc_cross_block.py > $OUT/pmc_c_timing.log 2>&1 || exit 1
  reduce || exit 1
  cat $OUT/bench_default.json | cut -c1-400 ;;
gemm)
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
  reduce || exit 1 ;;
cross)
  for mode in cached absorbed; do
    timeout -k 10 300 $B --cross-attention $mode --new-tokens 32 --steps 12 > $OUT/sw_${mode}_s64_n32.json 2>/dev/null || exit 1
    timeout -k 10 300 $B --cross-attention $mode --new-tokens 128 --steps 8 > $OUT/sw_${mode}_s64_n128.json 2>/dev/null || exit 1
    timeout -k 10 400 $B --cross-attention $mode --batch 128 --steps 6 > $OUT/sw_${mode}_s128_n64.json 2>/dev/null || exit 1
    timeout -k 10 400 $B --cross-attention $mode --batch 128 --new-tokens 224 --steps 4 > $OUT/sw_${mode}_s128_n224.json 2>/dev/null || exit 1
    timeout -k 10 500 $B --cross-attention $mode --model medium --batch 256 --pipeline 2 --new-tokens 224 --steps 3 > $OUT/sw_${mode}_m256_n224.json 2>/dev/null || exit 1
  done
  CROSS_TABLE=1 reduce || exit 1 ;;
train-size)
  echo "== fine-tune step"
  python3 $ROOT/bench.py --mode train --steps 5 --warmup 1 --f32 exact > $OUT/train_exact.json 2> $OUT/train_exact.err || exit 1
  python3 $ROOT/bench.py --mode train --steps 5 --warmup 1 --f32 split > $OUT/train_split.json 2> $OUT/train_split.err || exit 1
  echo "== medium B=256";          $B --model medium --batch 256 --pipeline 2 --steps 4 > $OUT/size_medium_b256.json 2> /dev/null || exit 1
  echo "== medium B=256 cached";   $B --model medium --batch 256 --pipeline 2 --steps 4 --cross-attention cached > $OUT/size_medium_b256_cached.json 2> /dev/null || exit 1
  echo "== large-v3 B=128 bf16";   $B --model large-v3 --batch 128 --pipeline 2 --steps 4 > $OUT/size_large_b128_bf16.json 2> /dev/null || exit 1
  echo "== large-v3 B=128 fp8 w";  $B --model large-v3 --batch 128 --pipeline 2 --steps 4 --weights fp8 > $OUT/size_large_b128_fp8.json 2> /dev/null || exit 1
  echo "== large-v3 B=128 fp8 w+a"; $B --model large-v3 --batch 128 --pipeline 2 --steps 4 --weights fp8 --activations fp8 > $OUT/size_large_b128_fp8_act.json 2> /dev/null || exit 1
  echo "== small fp8";             $B --weights fp8 > $OUT/size_small_fp8.json 2> /dev/null || exit 1
  echo "== small pipeline 1";      $B --pipeline 1 --steps 4 > $OUT/size_small_p1.json 2> /dev/null || exit 1
  for f in $OUT/size_*.json $OUT/train_*.json; do echo "$(basename $f): $(python3 -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['value'], d.get('decode_step',{}).get('ms_per_step'), d.get('roofline',{}).get('frac'), d.get('roofline_mfma',{}).get('frac'))")"; done
  reduce || exit 1 ;;
gaps)
  for BATCH in 8 64; do for P in 4 1; do
    timeout -k 10 200 python3 $ROOT/bench.py --phase dec --batch $BATCH --pipeline $P --steps 8 --warmup 1 2>/dev/null | tail -1 | tee -a $OUT/gaps_untraced.txt || exit 1
  done; done
  for P in 4 1; do
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/gaps_p$P -o t -- python3 $ROOT/bench.py --phase dec --batch 8 --pipeline $P --steps 8 --warmup 1 > $OUT/gaps_p$P.log 2>&1 || exit 1
    grep -h diagnostic_phase $OUT/gaps_p$P.log
  done
  python3 $ROOT/tools/decode_gaps.py $(find $OUT/gaps_p4 -name "*kernel_trace.csv" | head -1) $(find $OUT/gaps_p1 -name "*kernel_trace.csv" | head -1) | tee $KEEP/decode_gaps_table.txt
  cp $OUT/gaps_untraced.txt $KEEP/ ;;
chain-probe)
  hipcc --offload-arch=gfx950 -O3 -o /tmp/chain_probe $ROOT/tools/micro/chain_probe.hip 2>/dev/null || exit 1
  timeout -k 10 240 /tmp/chain_probe 48 | tee $KEEP/chain_probe.txt ;;
chain-pad)
  hipcc --offload-arch=gfx950 -O3 -o /tmp/chain_probe $ROOT/tools/micro/chain_probe.hip 2>/dev/null || exit 1
  for Q in 8 16 4; do for PAD in 0 1 2 3; do
    GPU_MAX_HW_QUEUES=$Q CHAIN_PAD=$PAD CHAIN_QUICK=1 timeout -k 10 60 /tmp/chain_probe 32 | cut -c1-190 | tee -a $KEEP/chain_pad.txt || exit 1
  done; done
  GPU_MAX_HW_QUEUES=8 CHAIN_PRIO=1 CHAIN_QUICK=1 timeout -k 10 60 /tmp/chain_probe 32 | cut -c1-190 | tee -a $KEEP/chain_pad.txt
  echo "== the decode loops themselves (8 clips, 4 passes in flight): torch's pool streams, then the library's own streams" | tee -a $KEEP/queue_pad_sweep.txt
  cd $ROOT && bash tools/queue_pad_sweep.sh 2>&1 | tee -a $KEEP/queue_pad_sweep.txt ;;
gemm-probe)
  hipcc --offload-arch=gfx950 -O3 -o /tmp/gemm_loop_probe $ROOT/tools/micro/gemm_loop_probe.hip 2>/dev/null || exit 1
  timeout -k 10 200 /tmp/gemm_loop_probe | tee $KEEP/gemm_loop_probe.txt || exit 1
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $KEEP/glp_pmc_a -o t -- /tmp/gemm_loop_probe > $KEEP/glp_pmc_a.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $KEEP/glp_pmc_b -o t -- /tmp/gemm_loop_probe > $KEEP/glp_pmc_b.log 2>&1 || exit 1 ;;  # ~170 KB of counter CSVs
group-sweep)
  timeout -k 10 600 python3 $ROOT/tools/group_sweep.py 24 2>/dev/null | tee $KEEP/group_sweep.txt ;;
final)
  cd $ROOT
  timeout -k 10 1000 python -m pytest tests -m gpu -q > $KEEP/final_tests.log 2>&1; tail -3 $KEEP/final_tests.log
  timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $KEEP/final_smoke.log 2>&1; tail -1 $KEEP/final_smoke.log
  S=$(date +%s); python bench.py > $KEEP/final_bench.json 2> $KEEP/final_bench.err; echo "bench rc=$? wall=$(( $(date +%s) - S )) s"
  python3 -c "
import json; d=json.loads(open('$KEEP/final_bench.json').read().strip().splitlines()[-1])
print('ms/pass', d['ms_per_step'], 'value', d['value'], 'single', d['ms_per_pass_single_in_flight'], 'evaluate-style', d['evaluate_style']['frac_of_value'],
      'step', d['decode_step']['ms_per_step'], 'roofline', d['roofline']['frac'], 'mfma', d['roofline_mfma']['frac'],
      'parity', d['parity_vs_cpu']['token_match'], d['parity_vs_cpu_peaky']['rows_identical'], 'other', {k: v['ms_per_step'] for k, v in d.get('other_configs', {}).items()})" ;;
*) echo "usage: bash tools/profile.sh bench|gemm|cross|train-size|gaps|chain-probe|chain-pad|gemm-probe|group-sweep|final   (ROUND=r05 SPLITS=2)"; exit 2 ;;
esac
echo "== done $MODE"
