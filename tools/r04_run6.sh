#!/bin/bash
# round 4, GPU session 6 (final tree): full tests, smoke, the driver's bench command
O=gpurun_out/r4s6; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
