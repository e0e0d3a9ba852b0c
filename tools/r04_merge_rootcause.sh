#!/bin/bash
# root-cause runs of the merge kernel's single-thread section (round 4); each python process is bounded.
# Needs the diagnostic build: WIPA_EXTRA_HIPCC_FLAGS=-DWIPA_MERGE_VARIANTS python -c "import __graft_entry__ as g; g.build(force=True)"
# (the shipped library has the every-lane kernel only and ignores WIPA_MERGE_SINGLE).
O=gpurun_out/r4_rootcause; mkdir -p $O
WIPA_MERGE_SINGLE=0 DIAG_SAVE=$O/base.pt timeout -k 10 120 python tools/merge_single_diag.py > $O/v0.log 2>&1
for v in 2 3 4; do
  WIPA_MERGE_SINGLE=$v DIAG_BASE=$O/base.pt timeout -k 10 120 python tools/merge_single_diag.py > $O/v${v}_4streams.log 2>&1
  WIPA_MERGE_SINGLE=$v DIAG_BASE=$O/base.pt DIAG_STREAMS=1 timeout -k 10 120 python tools/merge_single_diag.py > $O/v${v}_1stream.log 2>&1
done
for f in $O/*.log; do echo "== $f"; grep -v "^   clip" $f | cut -c1-220 | head -12; done
