#!/usr/bin/env python3
"""Small target for `rocprofv3 --pmc ...`: the dominant decode kernel (cross-attention, K11) on the
bench's shapes (whisper-small, B=64, bf16), 24 launches cycling through 12 layer caches so every
launch streams bytes that are not cache resident.  Whole-bench PMC runs are too slow (the counters
serialise ~18 000 dispatches), so HBM traffic is measured here and quoted per launch.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python tools/pmc_cross_attn.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -- python tools/pmc_cross_attn.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_ipa_amd import ops  # noqa: E402

B, H, T, L = 64, 12, 1500, 12
g = torch.Generator(device="cuda").manual_seed(0)
kv = (torch.randn(L, B, 2 * H, T, 64, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
q = (torch.randn(B, H * 64, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
for i in range(24):
    out = ops.decode_cross_attn(q, kv[i % L])
torch.cuda.synchronize()
print("algorithmic bytes per launch:", B * 2 * H * T * 64 * 2 + 2 * B * H * 64 * 2)
