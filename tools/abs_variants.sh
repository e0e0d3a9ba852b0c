cd $GRAFT_REPO_ROOT
for m in 0 1 2 3 4; do
  echo "== dbg $m"; WIPA_ABS_DBG=$m WIPA_ABS_STAGES=${STG:-7} timeout -k 10 120 python tools/cross_absorbed_bench.py 64 2>&1 | tail -n 2
done
