#!/usr/bin/env python3
"""Encoder flash attention in float32 (B = 32, H = 12, T = 1500): time and error against float64 on one (b, h), in the
three-term bf16 split mode (opt-in, f32_split=1) and with exact f32 products (default)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_ipa_amd import _lib, ops  # noqa: E402
from whisper_ipa_amd.runtime import stream  # noqa: E402

B, H, T = 32, 12, 1500
D = H * 64
g = torch.Generator(device="cuda").manual_seed(0)
qk = torch.randn(B * T, 2 * D, device="cuda", generator=g) * 0.6
v = torch.randn(B * T, D, device="cuda", generator=g)
b, h = 3, 5
q1 = qk.view(B, T, 2 * D)[b, :, h * 64:(h + 1) * 64].double()
k1 = qk.view(B, T, 2 * D)[b, :, D + h * 64:D + (h + 1) * 64].double()
v1 = v.view(B, T, D)[b, :, h * 64:(h + 1) * 64].double()
ref = torch.softmax(q1 @ k1.t(), dim=-1) @ v1
for mode in ("split", "exact"):
    split = mode == "split"
    out = ops.flash_attn_enc_f32(qk, v, B, H, T, f32_split=split)
    err = (out.view(B, T, D)[b, :, h * 64:(h + 1) * 64].double() - ref).abs().max().item()
    s = stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        e0.record(s)
        for _ in range(4):
            ops.flash_attn_enc_f32(qk, v, B, H, T, f32_split=split)
        e1.record(s)
    e1.synchronize()
    us = e0.elapsed_time(e1) / 4 * 1e3
    print(f"flash f32 {mode:5s}: {us:8.1f} us  {4.0 * B * H * T * T * 64 / us / 1e6:7.1f} TF/s  max abs err {err:.3e} (max |ref| {ref.abs().max().item():.2f})", flush=True)
