#!/usr/bin/env python3
"""f32 GEMM (the dtype the reference's scripts set) on the encoder shapes: time and error against float64.
WIPA_F32_GEMM=split takes every product as three bf16 MFMA terms (desc.f32_split); the default is the f32 MFMA."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_ipa_amd import ops  # noqa: E402
from whisper_ipa_amd.runtime import stream  # noqa: E402

M = 48000
SPLIT = os.environ.get("WIPA_F32_GEMM", "exact") == "split"
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K in (("qk", 1536, 768), ("mlp1", 3072, 768), ("mlp2", 768, 3072)):
    A = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) * 0.05
    out = torch.zeros(M, N, device="cuda")
    for _ in range(2):
        ops.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, f32_split=SPLIT)
    s = stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        e0.record(s)
        for _ in range(5):
            ops.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, f32_split=SPLIT)
        e1.record(s)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    ref = A[:512].double() @ W.double().t()
    err = (out[:512].double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    print(f"{('bf16x3' if SPLIT else 'exact'):7s} f32 {name:5s}: {ms * 1e3:8.1f} us {2.0 * M * N * K / (ms * 1e-3) / 1e12:7.1f} TF/s  "
          f"max abs err {err:.3e} (max |ref| {scale:.2f}, rel {err / scale:.2e})", flush=True)
