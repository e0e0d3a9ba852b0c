#!/usr/bin/env python3
"""Timing of the absorbed-projection cross-attention (wipa_cross_absorbed_attention: absorb-q + streaming + merge/project)
against the cached-K/V cross block at the bench shape (whisper-small, 64 clips, bf16).  `hot`: the same xa every launch (147 MB:
resident in the 256 MB Infinity Cache); `cold`: launches rotate over N_BUF different xa buffers (what several passes in flight
look like to the memory system).  usage: python tools/cross_absorbed_bench.py [B]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_ipa_amd import _lib  # noqa: E402
from whisper_ipa_amd.runtime import on_stream, ptr, sptr  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, Tk, N_BUF = 12, 1500, 5
d = H * 64
L = _lib.lib()
_lib.check(L.wipa_cross_absorbed_init(d))
g = torch.Generator(device="cuda").manual_seed(0)
with on_stream() as s:
    xas = [torch.randn(B, Tk, d, device="cuda", generator=g).bfloat16() for _ in range(N_BUF)]
    q = (torch.randn(B, d, device="cuda", generator=g) * 0.3).bfloat16()
    wkT = [(torch.randn(d, d, device="cuda", generator=g) * 0.05).bfloat16() for _ in range(12)]
    wv = [(torch.randn(d, d, device="cuda", generator=g) * 0.05).bfloat16() for _ in range(12)]
    bv = torch.zeros(d, device="cuda")
    out = torch.empty(B, d, device="cuda", dtype=torch.bfloat16)
    nbytes = L.wipa_cross_absorbed_scratch_bytes(B, d, Tk)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")

    def launch(i, rotate):
        xa = xas[i % N_BUF] if rotate else xas[0]
        _lib.check(L.wipa_cross_absorbed_attention(ptr(q), d, ptr(wkT[i % 12]), ptr(xa), ptr(wv[i % 12]), ptr(bv), ptr(out), d, ptr(scratch),
                                                   nbytes, B, H, d, Tk, 64 ** -0.25, 0, sptr(s)))

    for rotate in (False, True):
        for i in range(12):
            launch(i, rotate)
        s.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            for i in range(48):
                launch(i, rotate)
        graph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        graph.replay()
        e1.record(s)
        e1.synchronize()
        us = e0.elapsed_time(e1) / 48 * 1e3
        xa_bytes = B * Tk * d * 2
        print(f"B={B} absorbed cross-attention, {'cold (rotating xa)' if rotate else 'hot (one xa)'}: {us:.2f} us per layer call (3 launches) -> "
              f"{xa_bytes / us / 1e6:.2f} TB/s of xa; splits {L.wipa_cross_absorbed_splits(0, Tk)}", flush=True)
