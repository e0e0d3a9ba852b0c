#!/usr/bin/env python3
"""Calibration only (not a product path): the vendor GEMM library (torch.nn.functional.linear -> hipBLASLt / rocBLAS) on the
encoder GEMM shapes, bf16, random data -- what a tuned plain GEMM reaches on this device for K = 768 / 3072."""
import torch
M = 96000
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K in (("qk", 1536, 768), ("mlp1", 3072, 768), ("out", 768, 768), ("mlp2", 768, 3072), ("sq", 8192, 8192)):
    m = M if name != "sq" else 8192
    A = torch.randn(m, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    for _ in range(3):
        C = torch.nn.functional.linear(A, W)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        C = torch.nn.functional.linear(A, W)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"library bf16 GEMM {name:5s} M={m} N={N} K={K}: {ms*1e3:8.1f} us {2.0*m*N*K/(ms*1e-3)/1e12:7.1f} TF/s", flush=True)
