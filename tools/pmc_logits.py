#!/usr/bin/env python3
"""PMC target (no hipGraph: counters + graph replay hung once): the logits projection of a decode step,
M = 64 rows, N = 51 865, K = 768, bf16 -> f32, ten plain launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_ipa_amd import ops  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
M, N, K = 64, 51865, 768
A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
out = torch.zeros(M, 51872, device="cuda")
for _ in range(10):
    ops.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldw=K, ldc=51872)
torch.cuda.synchronize()
print("done")
