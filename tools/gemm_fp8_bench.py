#!/usr/bin/env python3
"""Micro-benchmark of the fp8 x fp8 tile GEMM (v_mfma_f32_16x16x128_f8f6f4) against the bf16 tile GEMM on the encoder
shapes of whisper-large-v3 (d = 1280) or whisper-small (d = 768), 64 clips.  usage: python tools/gemm_fp8_bench.py [small|large]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_ipa_amd import ops  # noqa: E402
from whisper_ipa_amd.runtime import stream  # noqa: E402
from whisper_ipa_amd.whisper import quantize_fp8_e4m3  # noqa: E402

d = 768 if (len(sys.argv) > 1 and sys.argv[1] == "small") else 1280
M = 96000
cases = [("qk   bias+scale -> bf16", 2 * d, d, torch.bfloat16, 0, False), ("mlp1 bias+gelu  -> bf16", 4 * d, d, torch.bfloat16, 1, False),
         ("mlp2 bias+resid -> f32 ", d, 4 * d, torch.float32, 0, True)]
g = torch.Generator(device="cuda").manual_seed(0)


def timed(fn):
    for _ in range(2):
        fn()
    s = stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        e0.record(s)
        for _ in range(5):
            fn()
        e1.record(s)
    e1.synchronize()
    return e0.elapsed_time(e1) / 5


for name, N, K, odt, act, resid in cases:
    A = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) * 0.05
    bias = torch.randn(N, device="cuda", generator=g)
    out = torch.zeros(M, N, device="cuda", dtype=odt)
    Ab, Wb = A.bfloat16(), W.bfloat16()
    ac, asc = quantize_fp8_e4m3(A)
    wc, wsc = quantize_fp8_e4m3(W)
    kw = dict(bias=bias, act=act, residual=out if resid else None, col_scale_n=0 if act else N, col_scale=0.35)
    ms_b = timed(lambda: ops.gemm(Ab, Wb, out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, **kw))
    ms_8 = timed(lambda: ops.gemm_fp8(ac, asc, wc, wsc, out, **kw))
    fl = 2.0 * M * N * K
    print(f"d={d} {name} N={N:5d} K={K:5d}: bf16 {ms_b * 1e3:7.1f} us = {fl / ms_b / 1e9:6.0f} TF/s | fp8 {ms_8 * 1e3:7.1f} us = {fl / ms_8 / 1e9:6.0f} TF/s "
          f"({ms_b / ms_8:.2f}x)", flush=True)
    del A, W, out, Ab, Wb, ac, wc
