#!/usr/bin/env python3
"""Do two streaming launches of the absorbed cross-attention on two HIP streams run side by side?  48 launches per stream, whisper-small
shapes (B = 64, d = 768, Tk = 1500), for SPLITS in (4, 2, 1): one stream alone, two streams with eager launches, two streams with one
captured graph each.  Prints wall time per launch of ONE stream's sequence (a perfect overlap of two streams = the single-stream time).  Every stream
reads its own two encoder outputs (12 launches each in turn), so nothing is shared between streams."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_ipa_amd import _lib  # noqa: E402
from whisper_ipa_amd.runtime import ptr  # noqa: E402

B, H, Tk, N = 64, 12, 1500, 48
d = H * 64
L = _lib.lib()
_lib.check(L.wipa_cross_absorbed_init(d))
g = torch.Generator(device="cuda").manual_seed(0)
xas = [torch.randn(B, Tk, d, device="cuda", generator=g).bfloat16() for _ in range(8)]  # two private encoder outputs per stream
q = (torch.randn(B, d, device="cuda", generator=g) * 0.3).bfloat16()
wkT = (torch.randn(d, d, device="cuda", generator=g) * 0.05).bfloat16()
wv = (torch.randn(d, d, device="cuda", generator=g) * 0.05).bfloat16()
bv = torch.zeros(d, device="cuda")
out = torch.empty(B, d, device="cuda", dtype=torch.bfloat16)
nbytes = L.wipa_cross_absorbed_scratch_bytes(B, d, Tk)
streams = [torch.cuda.Stream() for _ in range(4)]
scr = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(4)]
for i in range(4):
    _lib.check(L.wipa_cross_absorbed_attention(ptr(q), d, ptr(wkT), ptr(xas[0]), ptr(wv), ptr(bv), ptr(out), d, ptr(scr[i]), nbytes, B, H, d, Tk,
                                               64 ** -0.25, 0, streams[i].cuda_stream))
torch.cuda.synchronize()


def seq(i, splits):
    for k in range(N):
        _lib.check(L.wipa_cross_absorbed_stream(ptr(xas[2 * i + (k // 12) % 2]), ptr(scr[i]), nbytes, B, H, d, Tk, splits, streams[i].cuda_stream))


def wall(fn, n_streams):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(n_streams)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best / N * 1e6


for splits in (4, 2, 1):
    graphs = []
    for i in range(4):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=streams[i]):
            seq(i, splits)
        graphs.append(gr)

    def eager(n):
        for i in range(n):
            seq(i, splits)

    def replay(n):
        for i in range(n):
            with torch.cuda.stream(streams[i]):
                graphs[i].replay()

    print(f"splits {splits}: graph x1 {wall(replay, 1):6.1f}  x2 {wall(replay, 2):6.1f}  x4 {wall(replay, 4):6.1f} us | "
          f"eager x1 {wall(eager, 1):6.1f}  x2 {wall(eager, 2):6.1f}  x4 {wall(eager, 4):6.1f} us   (wall per launch of one stream's sequence)", flush=True)
