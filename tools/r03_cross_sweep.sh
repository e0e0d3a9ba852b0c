cd $GRAFT_REPO_ROOT
B="python3 bench.py --no-cpu-baseline --no-finetune"
for mode in cached absorbed; do
  timeout -k 10 300 $B --cross-attention $mode --new-tokens 128 --steps 8 > gpurun_out/sw_${mode}_s64_n128.json 2>/dev/null
  timeout -k 10 300 $B --cross-attention $mode --new-tokens 32 --steps 12 > gpurun_out/sw_${mode}_s64_n32.json 2>/dev/null
  timeout -k 10 400 $B --cross-attention $mode --batch 128 --new-tokens 224 --steps 4 > gpurun_out/sw_${mode}_s128_n224.json 2>/dev/null
  timeout -k 10 400 $B --cross-attention $mode --batch 128 --steps 6 > gpurun_out/sw_${mode}_s128_n64.json 2>/dev/null
  timeout -k 10 500 $B --cross-attention $mode --model medium --batch 256 --pipeline 2 --new-tokens 224 --steps 3 > gpurun_out/sw_${mode}_m256_n224.json 2>/dev/null
done
