#!/usr/bin/env python3
"""stream 0: one call at a time, checked after each; streams 1..3: background load of the same calls"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from whisper_ipa_amd import _lib
from whisper_ipa_amd.runtime import ptr
B, H, Tk = 64, 12, 1500
d = H * 64
L = _lib.lib()
_lib.check(L.wipa_cross_absorbed_init(d))
g = torch.Generator(device="cuda").manual_seed(0)
xa = torch.randn(B, Tk, d, device="cuda", generator=g).bfloat16()
q = (torch.randn(B, d, device="cuda", generator=g) * 0.3).bfloat16()
wkT = (torch.randn(d, d, device="cuda", generator=g) * 0.05).bfloat16()
wv = (torch.randn(d, d, device="cuda", generator=g) * 0.05).bfloat16()
bv = torch.zeros(d, device="cuda")
nbytes = L.wipa_cross_absorbed_scratch_bytes(B, d, Tk)
streams = [torch.cuda.Stream() for _ in range(4)]
outs = [torch.empty(B, d, device="cuda", dtype=torch.bfloat16) for _ in range(4)]
scr = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(4)]
torch.cuda.synchronize()
def call(i):
    _lib.check(L.wipa_cross_absorbed_attention(ptr(q), d, ptr(wkT), ptr(xa), ptr(wv), ptr(bv), ptr(outs[i]), d, ptr(scr[i]), nbytes, B, H, d, Tk,
                                               64 ** -0.25, streams[i].cuda_stream))
for i in range(4):
    call(i)
torch.cuda.synchronize()
ref, ref_scr = outs[0].clone(), scr[0].clone()
S = L.wipa_cross_absorbed_splits(B, Tk)
qp_bytes = B * 16 * d * 2
bad_out = bad_part = 0
bg, fg = os.environ.get("BG_STAGES", "7"), os.environ.get("FG_STAGES", "7")
print("background stages", bg, "foreground stages", fg)
for it in range(150):
    os.environ["WIPA_ABS_STAGES"] = bg
    for i in range(1, 4):
        for _ in range(2):
            call(i)
    os.environ["WIPA_ABS_STAGES"] = fg
    call(0)
    streams[0].synchronize()
    po = int((scr[0][qp_bytes:nbytes - 1024] != ref_scr[qp_bytes:nbytes - 1024]).sum())
    oo = int((outs[0] != ref).sum())
    bad_part += po > 0
    bad_out += oo > 0
    if po or oo:
        print("iter", it, "partial byte diffs", po, "out diffs", oo, flush=True)
print("calls with wrong partials:", bad_part, " with wrong out:", bad_out, "of 150")
