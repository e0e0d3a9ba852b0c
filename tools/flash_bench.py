#!/usr/bin/env python3
"""Encoder flash attention (K5) on the bench shape: B = 64, H = 12, T = 1500, bf16.  Times 6 launches by HIP events and
checks the result against an f32 softmax(QK^T)V of the same bf16 inputs on one (b, h).  Also the PMC target:
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
      SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d out -- python3 tools/flash_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_ipa_amd import ops  # noqa: E402
from whisper_ipa_amd.runtime import stream  # noqa: E402

B, H, T = 64, 12, 1500
D = H * 64
g = torch.Generator(device="cuda").manual_seed(0)
qk = (torch.randn(B * T, 2 * D, device="cuda", generator=g) * 0.6).bfloat16()
vt = torch.zeros(B, D, 1536, device="cuda", dtype=torch.bfloat16)
vt[:, :, :T] = (torch.randn(B, D, T, device="cuda", generator=g)).bfloat16()
out = ops.flash_attn_enc(qk, vt, B, H, T)
b, h = 3, 5
q = qk.view(B, T, 2 * D)[b, :, h * 64:(h + 1) * 64].float()
k = qk.view(B, T, 2 * D)[b, :, D + h * 64:D + (h + 1) * 64].float()
v = vt[b, h * 64:(h + 1) * 64, :T].float().t()
ref = torch.softmax(q @ k.t(), dim=-1) @ v
err = (out.view(B, T, D)[b, :, h * 64:(h + 1) * 64].float() - ref).abs().max().item()
s = stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(s):
    e0.record(s)
    for _ in range(6):
        ops.flash_attn_enc(qk, vt, B, H, T)
    e1.record(s)
e1.synchronize()
us = e0.elapsed_time(e1) / 6 * 1e3
flops = 4.0 * B * H * T * T * 64
print(f"flash_enc B={B} H={H} T={T}: {us:8.1f} us  {flops / us / 1e6:7.1f} TF/s  max err vs f32 reference {err:.3e}", flush=True)
