ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "8 1" "8 2" "8 4" "32 1" "32 4" "64 4"; do set -- $cfg
  echo -n "batch $1: "; timeout -k 10 200 python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --phase dec --batch $1 --pipeline $2 --steps 16 2>/dev/null | tail -1
done
echo cached; for p in 1 2 4; do timeout -k 10 200 python3 $ROOT/bench.py --no-cpu-baseline --no-finetune --phase dec --cross-attention cached --pipeline $p --steps 16 2>/dev/null | tail -1; done
