#!/bin/bash
# final validation of the tree: GPU tests, smoke, the default bench line
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r4_t13.log 2>&1; tail -3 gpurun_out/r4_t13.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_smoke13.log 2>&1; tail -1 gpurun_out/r4_smoke13.log
python bench.py > gpurun_out/r4_bench13.json 2> gpurun_out/r4_bench13.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r4_bench13.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['ms_per_pass_single_in_flight'], d['decode_step']['ms_per_step'], d['default_splits']['decode_step_ms'], d['roofline']['frac'], d['roofline']['two_launches_side_by_side']['frac'], d['parity_vs_cpu']['token_match'], d['parity_vs_cpu_peaky']['rows_identical'])"
