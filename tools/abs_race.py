#!/usr/bin/env python3
"""Determinism probe of wipa_cross_absorbed_attention under concurrency: the same call on 4 HIP streams at once, repeated;
every output must equal the single-stream output bit for bit."""
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
call(0); torch.cuda.synchronize(); ref = outs[0].clone(); ref_scr = scr[0].clone()
S = L.wipa_cross_absorbed_splits(B, Tk)
qp_bytes = B * 16 * d * 2
pm = B * S * 16 * 4
for i in range(1, 4):
    call(i)
torch.cuda.synchronize()
if len(sys.argv) > 1:
    os.environ["WIPA_ABS_STAGES"] = sys.argv[1]
    print("stages mask", sys.argv[1])
for trial in range(8):
    for rep in range(20):
        for i in range(4):
            call(i)
    torch.cuda.synchronize()
    rep = []
    for i in range(4):
        o = int((outs[i] != ref).sum())
        qd = int((scr[i][:qp_bytes] != ref_scr[:qp_bytes]).sum())
        md = int((scr[i][qp_bytes:qp_bytes + 2 * pm] != ref_scr[qp_bytes:qp_bytes + 2 * pm]).sum())
        od = int((scr[i][qp_bytes + 2 * pm:nbytes - 1024] != ref_scr[qp_bytes + 2 * pm:nbytes - 1024]).sum())
        rep.append((o, qd, md, od))
        if o:
            bad = (outs[i] != ref).nonzero()
            rep.append(("rows", sorted(set(bad[:, 0].tolist()))[:4], "cols", int(bad[:, 1].min()), int(bad[:, 1].max())))
            r0, c0 = int(bad[0, 0]), int(bad[:, 1].min())
            print("   got", outs[i][r0, c0:c0 + 6].float().tolist(), "want", ref[r0, c0:c0 + 6].float().tolist(), "ratio",
                  (outs[i][r0, c0:c0 + 6].float() / ref[r0, c0:c0 + 6].float()).tolist())
    print("trial", trial, "(out, Qp, m/l, O') byte diffs per stream:", rep, flush=True)
