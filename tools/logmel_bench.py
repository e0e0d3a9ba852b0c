#!/usr/bin/env python3
"""Time of the log-mel front-end for a batch of 30 s clips (default 64), HIP events over N calls.
WIPA_LOGMEL=gemm selects the previous form (f32 STFT GEMM + mel kernel).  usage: python tools/logmel_bench.py [B] [n_mels]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_ipa_amd import audio  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_mels = int(sys.argv[2]) if len(sys.argv) > 2 else 80
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(B, 480000, device="cuda", generator=g) * 0.1
for _ in range(3):
    audio.log_mel_padded(a, n_mels, torch.bfloat16)
torch.cuda.synchronize()
N = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s = torch.cuda.current_stream()
e0.record(s)
for _ in range(N):
    audio.log_mel_padded(a, n_mels, torch.bfloat16)
e1.record(s)
e1.synchronize()
ms = e0.elapsed_time(e1) / N
alg = B * (480000 * 4 + 3000 * n_mels * 2)
print(f"log-mel B={B} n_mels={n_mels} form={os.environ.get('WIPA_LOGMEL', 'fused')}: {ms * 1e3:.1f} us per batch; audio in + bf16 mel out = {alg / 1e6:.1f} MB -> "
      f"{alg / ms / 1e6:.1f} GB/s")
