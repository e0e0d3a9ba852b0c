#!/usr/bin/env python3
"""Small target for `rocprofv3 --pmc ...`: the five encoder GEMMs of whisper-small at B=64 (M = 96000 rows),
with the epilogues the encoder uses, three launches each on random data.

  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
      --output-format csv -d out -- python3 tools/pmc_gemm.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_ipa_amd import ops  # noqa: E402

M = 96000
cases = [  # name, N, K, out dtype, act, residual, col_scale
    ("qk", 1536, 768, torch.bfloat16, 0, False, True),
    ("mlp1", 3072, 768, torch.bfloat16, 1, False, False),
    ("out", 768, 768, torch.float32, 0, True, False),
    ("mlp2", 768, 3072, torch.float32, 0, True, False),
    ("mlp2_plain", 768, 3072, torch.bfloat16, 0, False, False),
]
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K, odt, act, resid, scale in cases:
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    out = torch.zeros(M, N, device="cuda", dtype=odt)
    for _ in range(3):
        ops.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, bias=bias, act=act, residual=out if resid else None,
                 col_scale_n=N if scale else 0, col_scale=0.35)
    torch.cuda.synchronize()
    print(name, "done", flush=True)
