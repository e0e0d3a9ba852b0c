#!/usr/bin/env python3
"""Root-cause aid (round 4; needs a library built with WIPA_EXTRA_HIPCC_FLAGS=-DWIPA_MERGE_VARIANTS), WIPA_MERGE_SINGLE=5: the merge kernel's single-thread section DUMPS, per (head, clip): the split weights
as a lane of another wave read them from LDS, the same LDS words re-read by thread 0, and the (m, l) statistics thread 0 loaded.
The host recomputes the weights from the statistics the STREAMING kernel left in the scratch and says which stage is off."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["WIPA_MERGE_SINGLE"] = "5"
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
torch.cuda.synchronize()
n_bad = 0
kinds = {"loaded_stats": 0, "computed_weights": 0, "readback_other_wave": 0, "readback_thread0": 0}
for it in range(int(os.environ.get("DUMP_ITERS", "200"))):
    _lib.check(L.wipa_cross_absorbed_attention(ptr(q), d, ptr(wkT), ptr(xa), ptr(wv), ptr(bv), ptr(out), d, ptr(scr), nbytes, B, H, d, Tk,
                                               64 ** -0.25, 0, st.cuda_stream))
    torch.cuda.synchronize()
    dump = out.view(torch.int16).cpu().numpy().view(np.float32).reshape(B, H, 32)  # 64 bf16 = 32 floats per (clip, head)
    stats = scr[B * 16 * d * 2:].cpu().numpy().view(np.float32)
    pm = stats[: B * S * 16].reshape(B, S, 16)[:, :, :H].transpose(0, 2, 1)            # [B, H, S]
    pl = stats[B * S * 16: 2 * B * S * 16].reshape(B, S, 16)[:, :, :H].transpose(0, 2, 1)
    w_other, w_t0, m_t0, l_t0 = dump[:, :, 0:4], dump[:, :, 4:8], dump[:, :, 8:12], dump[:, :, 12:16]
    M = pm.max(-1, keepdims=True)
    e = np.exp((pm - M).astype(np.float64))
    want = e / (e * pl).sum(-1, keepdims=True)
    bad_stats = (m_t0 != pm) | (l_t0 != pl)
    bad_w = np.abs(w_t0 - want) > 1e-5 * np.abs(want)
    bad_rb = w_other != w_t0
    if bad_stats.any() or bad_w.any() or bad_rb.any():
        n_bad += 1
        for name, arr in (("loaded_stats", bad_stats), ("computed_weights", bad_w & ~bad_stats.any(-1, keepdims=True)), ("readback_other_wave", bad_rb)):
            if arr.any():
                kinds[name] += 1
                bb, hh = np.nonzero(arr.any(-1))
                b0, h0 = int(bb[0]), int(hh[0])
                print(f"launch {it}: {name} wrong in {len(bb)} (clip, head) pairs, e.g. clip {b0} head {h0}: loaded m {m_t0[b0, h0]} scratch m {pm[b0, h0]} | "
                      f"loaded l {l_t0[b0, h0]} scratch l {pl[b0, h0]} | weights thread0 {w_t0[b0, h0]} other wave {w_other[b0, h0]} want {want[b0, h0].astype(np.float32)}")
print("launches with any discrepancy:", n_bad, kinds)
