ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for m in 1 0 1 0; do echo -n "WIPA_LOGITS_FUSED=$m dec-only 4 in flight: "; WIPA_LOGITS_FUSED=$m timeout -k 10 200 python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --phase dec --pipeline 4 --steps 24 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_pass'])"; done
