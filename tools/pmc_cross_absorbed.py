#!/usr/bin/env python3
"""PMC / timing target: the dominant kernel of the default decode step in round 3, the streaming kernel of the absorbed
cross-attention (cross_absorbed_v2_kernel: one pass over the encoder output per layer), on the bench's shapes (whisper-small,
B = 64, bf16), 48 launches in the decode loop's order: 12 consecutive launches (the layers of a step) per encoder output, 4
encoder outputs (passes in flight) in turn (NBUF / PER_BUF in the environment change the pattern: NBUF=5 PER_BUF=1 is all-cold).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/f -- python3 tools/pmc_cross_absorbed.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/w -- python3 tools/pmc_cross_absorbed.py
Without counters it also prints the event-timed average per launch (graph replay).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_ipa_amd import _lib  # noqa: E402
from whisper_ipa_amd.runtime import on_stream, ptr, sptr  # noqa: E402

B, H, Tk = 64, 12, 1500
NBUF = int(os.environ.get("NBUF", "4"))
PER_BUF = int(os.environ.get("PER_BUF", "12"))
SPLITS = int(os.environ.get("SPLITS", "0"))  # frame splits per clip of the streaming launch: 0 = the default (4); 2 = bench.py's pipelined setting
d = H * 64
L = _lib.lib()
_lib.check(L.wipa_cross_absorbed_init(d))
g = torch.Generator(device="cuda").manual_seed(0)
with on_stream() as s:
    xas = [torch.randn(B, Tk, d, device="cuda", generator=g).bfloat16() for _ in range(NBUF)]
    q = (torch.randn(B, d, device="cuda", generator=g) * 0.3).bfloat16()
    wkT = (torch.randn(d, d, device="cuda", generator=g) * 0.05).bfloat16()
    wv = (torch.randn(d, d, device="cuda", generator=g) * 0.05).bfloat16()
    bv = torch.zeros(d, device="cuda")
    out = torch.empty(B, d, device="cuda", dtype=torch.bfloat16)
    nbytes = L.wipa_cross_absorbed_scratch_bytes(B, d, Tk)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    # one whole call fills the absorbed queries the streaming kernel reads
    _lib.check(L.wipa_cross_absorbed_attention(ptr(q), d, ptr(wkT), ptr(xas[0]), ptr(wv), ptr(bv), ptr(out), d, ptr(scratch), nbytes, B, H, d, Tk,
                                               64 ** -0.25, 0, sptr(s)))

    def launch(i):
        _lib.check(L.wipa_cross_absorbed_stream(ptr(xas[(i // PER_BUF) % NBUF]), ptr(scratch), nbytes, B, H, d, Tk, SPLITS, sptr(s)))

    for i in range(48):
        launch(i)
    s.synchronize()
    S = L.wipa_cross_absorbed_splits(SPLITS, Tk)
    xa_bytes = B * Tk * d * 2
    alg = xa_bytes + B * 16 * d * 2 + B * S * (H * d + 32) * 4
    print("xa bytes per launch:", xa_bytes, " algorithmic bytes per launch (xa once + absorbed queries in + split partials out):", alg)
    if "ROCPROF_COUNTER_COLLECTION" not in os.environ:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            for i in range(48):
                launch(i)
        graph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        graph.replay()
        e1.record(s)
        e1.synchronize()
        us = e0.elapsed_time(e1) / 48 * 1e3
        print(f"cross_absorbed_v2_kernel: {us:.2f} us per launch -> {alg / us / 1e6:.2f} TB/s")
