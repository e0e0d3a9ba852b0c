#!/bin/bash
# round 4, GPU session 5: full tests (incl. the fp8 384 x 256 tile in a subprocess) + fp8 tile A/B on the large-v3 shapes and the large-v3 pass
O=gpurun_out/r4s5; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for t in 256 384; do echo "== WIPA_GEMM_FP8_TILE=$t"; WIPA_GEMM_FP8_TILE=$t python tools/gemm_fp8_bench.py large 2>&1 | grep "d="; done | tee $O/fp8_tile_ab.log
for t in 256 0; do WIPA_GEMM_FP8_TILE=$t python bench.py --no-cpu-baseline --no-finetune --model large-v3 --batch 128 --pipeline 2 --steps 4 --weights fp8 --activations fp8 > $O/large_fp8_act_tile$t.json 2> /dev/null; done
python - <<PY
import json
for t in (256, 0):
    d = json.loads(open("$O/large_fp8_act_tile%d.json" % t).read().strip().splitlines()[-1])
    print("large-v3 B=128 fp8 w+a, WIPA_GEMM_FP8_TILE=%d:" % t, d["ms_per_step"], d["value"], d.get("roofline_mfma", {}).get("frac"), d.get("roofline_mfma", {}).get("gemm_ms_per_pass"), d["passes_identical"])
PY
