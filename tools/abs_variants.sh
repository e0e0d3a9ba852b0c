cd /tmp && export TMPDIR=/tmp
for st in 1 2 4 5 6; do
  WIPA_ABS_STAGES=$st timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_st$st -o st -- python3 $GRAFT_REPO_ROOT/tools/cross_absorbed_bench.py 64 > $GRAFT_REPO_ROOT/gpurun_out/prof_st$st.log 2>&1 || exit 1
done
