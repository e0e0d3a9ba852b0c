#!/usr/bin/env python3
"""PMC / timing target: the dominant kernel of the default decode step, wipa_decode_cross_block (split-K slab sum + residual +
cross_attn_ln + cross query + streaming cross-attention in one launch), on the bench's shapes (whisper-small, B = 64, bf16),
24 launches cycling through 12 layer caches so every launch streams K/V bytes that are not cache resident.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/f -- python3 tools/pmc_cross_block.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/w -- python3 tools/pmc_cross_block.py
Without counters it prints the event-timed average per launch.
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_ipa_amd import _lib  # noqa: E402
from whisper_ipa_amd.runtime import on_stream, ptr, sptr, stream  # noqa: E402

B, H, T, L = 64, 12, 1500, 12
d = H * 64
g = torch.Generator(device="cuda").manual_seed(0)
lib = _lib.lib()
with on_stream() as s:
    kv = (torch.randn(L, B, 2 * H, T, 64, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    x = torch.randn(B, d, device="cuda", generator=g)
    x_out = torch.empty_like(x)
    slabs = torch.randn(2, B, d, device="cuda", generator=g) * 0.3
    ln_w, ln_b = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
    wq = [(torch.randn(d, d, device="cuda", generator=g) * 0.05).to(torch.bfloat16) for _ in range(L)]
    bq = torch.zeros(d, device="cuda")
    out = torch.empty(B, d, device="cuda", dtype=torch.bfloat16)

    def launch(i):
        c = _lib.CrossBlockDesc()
        c.x_in, c.x_out, c.slabs, c.bias_o, c.ln_w, c.ln_b = ptr(x), ptr(x_out), ptr(slabs), None, ptr(ln_w), ptr(ln_b)
        c.wq, c.bq, c.kv, c.out = ptr(wq[i % L]), ptr(bq), ptr(kv[i % L]), ptr(out)
        c.slab_stride = B * d
        c.n_slabs, c.B, c.d, c.H, c.Tk, c.dtype, c.eps, c.qk_scale = 2, B, d, H, T, _lib.WIPA_BF16, 1e-5, 64 ** -0.25
        _lib.check(lib.wipa_decode_cross_block(C.byref(c), sptr(s)), "wipa_decode_cross_block")

    for i in range(24):
        launch(i)
    s.synchronize()
    kv_bytes = B * 2 * H * T * 64 * 2
    alg = kv_bytes + B * d * (4 + 2 * 4 + 4 + 2) + d * d * 2  # + residual row in/out, two slab rows, output row, the query weights once
    print("K/V bytes per launch:", kv_bytes, " algorithmic bytes per launch (K/V + rows + query weights once):", alg)
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
        print(f"wipa_decode_cross_block: {us:.2f} us per launch -> {alg / us / 1e6:.2f} TB/s")
