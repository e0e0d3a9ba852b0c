#!/bin/bash
# round 4, GPU session 4: full tests + A/B of the decode-step knobs + flash softmax variant (one call: acquiring a box is charged)
O=gpurun_out/r4s4; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
B="python bench.py --no-finetune --no-cpu-baseline"
$B > $O/b_default.json 2> $O/b_default.err
WIPA_SELF_ATTN_WAVES=1 $B > $O/b_selfattn1.json 2> /dev/null
WIPA_ABS_PROLOGUE_CLIPS=16 $B > $O/b_prologue16.json 2> /dev/null
WIPA_DECODE_TAIL=0 $B > $O/b_notail.json 2> /dev/null
$B --new-tokens 224 --steps 6 > $O/b_n224_auto.json 2> /dev/null
$B --new-tokens 224 --steps 6 --cross-attention absorbed > $O/b_n224_absorbed.json 2> /dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/b_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["ms_per_pass_single_in_flight"], d["decode_step"]["ms_per_step"], d["passes_identical"], d["roofline"].get("avg_launch_ms"), d["config"]["cross_attention"], d.get("roofline_mfma", {}).get("flash_attention"))
    except Exception as e:
        print(f, "FAILED", e)
PY
for pk in 1 0; do WIPA_FLASH_PK=$pk python tools/flash_bench.py 2>&1 | tail -1 | sed "s/^/PK=$pk: /"; WIPA_FLASH_PK=$pk python tools/flash_bench.py 2>&1 | tail -1 | sed "s/^/PK=$pk: /"; done | tee $O/flash_pk.log
