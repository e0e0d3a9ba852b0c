#!/usr/bin/env python3
"""Assemble DESIGN.md from its section files under tools/design_parts/ (one set of numbers per section; the history of rounds 1-3 and the
round-4 summary live in DESIGN_HISTORY.md).  usage: python tools/build_design.py"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PARTS = os.path.join(ROOT, "tools", "design_parts")
ORDER = ["00_head.md", "0_round5.md", "1_path.md", "2_parity.md", "3_layout.md", "4_kernels.md", "5_multigpu.md", "6_measurements.md",
         "7_status.md", "8_experiments.md"]

out = []
for name in ORDER:
    out.append(open(os.path.join(PARTS, name)).read().rstrip("\n"))
open(os.path.join(ROOT, "DESIGN.md"), "w").write("\n\n".join(out) + "\n")
print("DESIGN.md:", sum(len(p) for p in out), "bytes from", len(ORDER), "parts")
