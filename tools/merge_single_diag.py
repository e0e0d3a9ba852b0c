#!/usr/bin/env python3
"""Root-cause aid (round 4; needs a library built with WIPA_EXTRA_HIPCC_FLAGS=-DWIPA_MERGE_VARIANTS): the merge kernel's single-thread weight section (WIPA_MERGE_SINGLE=1 scalar loads / 2 vector loads)
under four streams in flight.  On a mismatch, say WHAT differs: the split partials the streaming kernel wrote (producer side) or
only the merge output, which (clip, head) blocks, and what the wrong values look like.
usage: WIPA_MERGE_SINGLE=1 python tools/merge_single_diag.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
NS = int(os.environ.get("DIAG_STREAMS", "4"))
streams = [torch.cuda.Stream() for _ in range(NS)]
outs = [torch.empty(B, d, device="cuda", dtype=torch.bfloat16) for _ in range(NS)]
scr = [torch.zeros(nbytes, dtype=torch.uint8, device="cuda") for _ in range(NS)]
torch.cuda.synchronize()


def call(i):
    _lib.check(L.wipa_cross_absorbed_attention(ptr(q), d, ptr(wkT), ptr(xa), ptr(wv), ptr(bv), ptr(outs[i]), d, ptr(scr[i]), nbytes, B, H, d,
                                               Tk, 64 ** -0.25, 0, streams[i].cuda_stream))


call(0)
torch.cuda.synchronize()
ref, ref_scr = outs[0].clone(), scr[0].clone()
print("variant WIPA_MERGE_SINGLE =", os.environ.get("WIPA_MERGE_SINGLE", "0"), "streams", NS)
save = os.environ.get("DIAG_SAVE")
if save:
    torch.save(ref.cpu(), save)
base = os.environ.get("DIAG_BASE")  # the default variant's output of the same inputs (bit reference)
base_t = torch.load(base).cuda() if base else None
if base_t is not None:
    d0 = (ref != base_t)
    print("first call vs the default variant's bits:", int(d0.sum()), "elements differ, blocks",
          sorted(set((int(r), int(c) // 64) for r, c in zip(*[t.tolist() for t in d0.nonzero(as_tuple=True)])))[:12])
print("NaN in first call:", bool(torch.isnan(ref.float()).any()))
bad = 0
for rep in range(6):
    for _ in range(12):
        for i in range(NS):
            call(i)
    torch.cuda.synchronize()
    for i in range(NS):
        eq_out = torch.equal(outs[i], ref)
        eq_scr = torch.equal(scr[i][: nbytes - 1024], ref_scr[: nbytes - 1024])
        if eq_out and eq_scr:
            continue
        bad += 1
        if torch.isnan(outs[i].float()).any():
            print(f"rep {rep} stream {i}: NaN in the output at", sorted(set((int(r), int(c) // 64) for r, c in zip(*[t.tolist() for t in torch.isnan(outs[i].float()).nonzero(as_tuple=True)])))[:8])
        if base_t is not None:
            db = (outs[i] != base_t)
            rel = ((outs[i].float() - base_t.float()).abs() / (base_t.float().abs() + 1e-3))[db]
            print(f"rep {rep} stream {i}: vs the default variant's bits {int(db.sum())} elements differ; relative size median {float(rel.median()) if rel.numel() else 0:.2e} max {float(rel.max()) if rel.numel() else 0:.2e}")
        diff = (outs[i] != ref)
        rows, cols = diff.nonzero(as_tuple=True)
        blocks = sorted(set((int(r), int(c) // 64) for r, c in zip(rows.tolist(), cols.tolist())))
        print(f"rep {rep} stream {i}: partials equal {eq_scr}; output differs in {int(diff.sum())} elements, (clip, head) blocks {blocks[:12]}")
        for (r, h) in blocks[:3]:
            a, b_ = outs[i][r, h * 64:(h + 1) * 64].float(), ref[r, h * 64:(h + 1) * 64].float()
            ratio = ((a - bv[h * 64:(h + 1) * 64]) / (b_ - bv[h * 64:(h + 1) * 64] + 1e-9))
            print(f"   clip {r} (cl = {r % 4}) head {h}: got[:4] {a[:4].tolist()} want[:4] {b_[:4].tolist()}  (got - bv)/(want - bv) median {float(ratio.median()):.4f} "
                  f"min {float(ratio.min()):.4f} max {float(ratio.max()):.4f}")
print("mismatching (rep, stream) pairs:", bad)
