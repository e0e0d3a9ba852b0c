#!/usr/bin/env python3
"""Root-cause aid (round 4; needs a library built with WIPA_EXTRA_HIPCC_FLAGS=-DWIPA_MERGE_VARIANTS), WIPA_MERGE_SINGLE=6: the FULL merge kernel with the single-thread section; every wave also stores the
split weights it read from LDS into never-read scratch rows.  For every launch whose output differs from the all-lanes kernel's
bits (DIAG_BASE, made by tools/merge_single_diag.py with WIPA_MERGE_SINGLE=0), say whether the weights of the differing
(clip, head) workgroups were right in every wave."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["WIPA_MERGE_SINGLE"] = "6"
from whisper_ipa_amd import _lib  # noqa: E402
from whisper_ipa_amd.runtime import ptr  # noqa: E402

L = _lib.lib()
B, H, Tk = 64, 12, 1500
d = H * 64
_lib.check(L.wipa_cross_absorbed_init(d))
g = torch.Generator(device="cuda").manual_seed(0)
xa = torch.randn(B, Tk, d, device="cuda", generator=g).bfloat16()
q = (torch.randn(B, d, device="cuda", generator=g) * 0.3).bfloat16()
wkT = (torch.randn(d, d, device="cuda", generator=g) * 0.05).bfloat16()
wv = (torch.randn(d, d, device="cuda", generator=g) * 0.05).bfloat16()
bv = torch.randn(d, device="cuda", generator=g) * 0.1
nbytes = L.wipa_cross_absorbed_scratch_bytes(B, d, Tk)
S = L.wipa_cross_absorbed_splits(0, Tk)
st = torch.cuda.Stream()
out = torch.zeros(B, d, device="cuda", dtype=torch.bfloat16)
scr = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
base = torch.load(os.environ["DIAG_BASE"]).cuda()
torch.cuda.synchronize()
n_glitch = n_w_bad = 0
for it in range(int(os.environ.get("DUMP_ITERS", "300"))):
    _lib.check(L.wipa_cross_absorbed_attention(ptr(q), d, ptr(wkT), ptr(xa), ptr(wv), ptr(bv), ptr(out), d, ptr(scr), nbytes, B, H, d, Tk,
                                               64 ** -0.25, 0, st.cuda_stream))
    torch.cuda.synchronize()
    diff = (out != base)
    stats = scr[B * 16 * d * 2:].cpu().numpy().view(np.float32)
    pm = stats[: B * S * 16].reshape(B, S, 16)[:, :, :H].transpose(0, 2, 1)
    pl = stats[B * S * 16: 2 * B * S * 16].reshape(B, S, 16)[:, :, :H].transpose(0, 2, 1)
    po = stats[2 * B * S * 16: 2 * B * S * 16 + B * S * 16 * d].reshape(B, S, 16, d)
    M = pm.max(-1, keepdims=True)
    e = np.exp((pm - M).astype(np.float64))
    want = (e / (e * pl).sum(-1, keepdims=True)).astype(np.float32)           # [B, H, S]
    got = np.stack([po[:, 0, 12 + (w >> 1), (w & 1) * 64: (w & 1) * 64 + 4 * H].reshape(B, H, 4) for w in range(8)], 0)  # [wave, B, H, S]
    w_bad = np.abs(got - want[None]) > 2e-6 * np.abs(want[None])
    if w_bad.any():
        n_w_bad += 1
        wv_, bb, hh, ss = np.nonzero(w_bad)
        print(f"launch {it}: weights wrong for {len(set(zip(bb.tolist(), hh.tolist())))} (clip, head) pairs, e.g. wave {wv_[0]} clip {bb[0]} head {hh[0]}: got {got[wv_[0], bb[0], hh[0]]} want {want[bb[0], hh[0]]}")
    if diff.any():
        n_glitch += 1
        rows, cols = diff.nonzero(as_tuple=True)
        blocks = sorted(set((int(r), int(c) // 64) for r, c in zip(rows.tolist(), cols.tolist())))
        wrong_here = [bool(w_bad[:, b_, h_].any()) for b_, h_ in blocks]
        print(f"launch {it}: output differs from the all-lanes kernel in blocks {blocks[:8]}; weights wrong in those blocks (any wave): {wrong_here[:8]}")
print(f"launches: output glitches {n_glitch}, weight discrepancies {n_w_bad}")
