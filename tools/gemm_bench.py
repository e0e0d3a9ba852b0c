#!/usr/bin/env python3
"""Micro-benchmark of wipa_gemm on the encoder shapes (run on the GPU box).
usage: python tools/gemm_bench.py [tag]   (WIPA_GEMM_TILE=128|256 selects the tile kernel)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_ipa_amd import ops  # noqa: E402
from whisper_ipa_amd.runtime import stream  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else ""
M = 96000
cases = [
    ("qk      N=1536 K=768  bf16->bf16 bias+scale", 1536, 768, torch.bfloat16, 0, False, True),
    ("mlp1    N=3072 K=768  bf16->bf16 bias+gelu ", 3072, 768, torch.bfloat16, 1, False, False),
    ("mlp1*   N=3072 K=768  bf16->bf16 bias      ", 3072, 768, torch.bfloat16, 0, False, False),
    ("out     N=768  K=768  bf16->f32  bias+resid", 768, 768, torch.float32, 0, True, False),
    ("out*    N=768  K=768  bf16->bf16 bias      ", 768, 768, torch.bfloat16, 0, False, False),
    ("mlp2    N=768  K=3072 bf16->f32  bias+resid", 768, 3072, torch.float32, 0, True, False),
    ("mlp2*   N=768  K=3072 bf16->bf16 bias      ", 768, 3072, torch.bfloat16, 0, False, False),
]
if len(sys.argv) > 2 and sys.argv[2] == "large":  # whisper-large-v3 width (d = 1280), 64 clips
    cases = [
        ("qk      N=2560 K=1280 bf16->bf16 bias+scale", 2560, 1280, torch.bfloat16, 0, False, True),
        ("mlp1    N=5120 K=1280 bf16->bf16 bias+gelu ", 5120, 1280, torch.bfloat16, 1, False, False),
        ("out     N=1280 K=1280 bf16->f32  bias+resid", 1280, 1280, torch.float32, 0, True, False),
        ("mlp2    N=1280 K=5120 bf16->f32  bias+resid", 1280, 5120, torch.float32, 0, True, False),
        ("mlp2*   N=1280 K=5120 bf16->bf16 bias      ", 1280, 5120, torch.bfloat16, 0, False, False),
    ]
if len(sys.argv) > 2 and sys.argv[2] == "medium":  # whisper-medium width (d = 1024), 64 clips
    cases = [
        ("qk      N=2048 K=1024 bf16->bf16 bias+scale", 2048, 1024, torch.bfloat16, 0, False, True),
        ("mlp1    N=4096 K=1024 bf16->bf16 bias+gelu ", 4096, 1024, torch.bfloat16, 1, False, False),
        ("out     N=1024 K=1024 bf16->f32  bias+resid", 1024, 1024, torch.float32, 0, True, False),
        ("mlp2    N=1024 K=4096 bf16->f32  bias+resid", 1024, 4096, torch.float32, 0, True, False),
        ("mlp2*   N=1024 K=4096 bf16->bf16 bias      ", 1024, 4096, torch.bfloat16, 0, False, False),
    ]
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K, odt, act, resid, scale in cases:
    A = (torch.randn(M, K, device="cuda", generator=g)).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    out = torch.zeros(M, N, device="cuda", dtype=odt)
    kw = dict(M=M, N=N, K=K, lda=K, ldw=K, ldc=N, bias=bias, act=act, residual=out if resid else None,
              col_scale_n=N if scale else 0, col_scale=0.35)
    for _ in range(2):
        ops.gemm(A, W, out, **kw)
    s = stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        e0.record(s)
        for _ in range(5):
            ops.gemm(A, W, out, **kw)
        e1.record(s)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    osz = 4 if odt == torch.float32 else 2
    gb = (M * K * 2 + N * K * 2 + M * N * osz * (2 if resid else 1)) / 1e9
    # correctness on a sample of rows (first / last tiles and 2000 random rows) against torch in float32
    out2 = torch.zeros(M, N, device="cuda", dtype=odt)
    ops.gemm(A, W, out2, **dict(kw, residual=out2 if resid else None))
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(0, 600), torch.arange(M - 600, M), torch.randint(0, M, (2000,))]).cuda()
    ref = A[rows].float() @ W.float().T + bias
    if scale:
        ref = ref * 0.35
    if act:
        ref = torch.nn.functional.gelu(ref)
    err = float((out2[rows].float() - ref).abs().max() / ref.abs().max())
    print(f"{tag:8s} {name}: {ms * 1e3:8.1f} us  {tf:7.1f} TF/s  min-traffic {gb:5.2f} GB -> {gb / (ms * 1e-3) / 1e3:5.2f} TB/s  err {err:.2e}", flush=True)
    del A, W, out, out2
