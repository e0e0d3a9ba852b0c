ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() { echo -n "$1 | batch $2 passes $3: "; env $1 timeout -k 10 200 python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --phase dec --batch $2 --pipeline $3 --steps 16 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_pass'])"; }
for b in 8 64; do for p in 1 4; do
run "X=0" $b $p
run "WIPA_ABS_FUSED_PROLOGUE=0" $b $p
run "WIPA_DECODE_TAIL=0" $b $p
run "WIPA_ABS_MERGE_OUT=1" $b $p
done; done
