#!/usr/bin/env python3
"""Do the decode step's small kernels (skinny projection, LayerNorm) make progress beside the absorbed cross-attention's
streaming kernel of ANOTHER HIP stream?  Same method as tools/coexec.py: each loop alone, then both at once.
WIPA_ABS_KERNEL=1 selects the channel-split streaming kernel (104 KiB of LDS, 4 waves) instead of the independent-wave one
(144 KiB, 3 waves x 415 registers)."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_ipa_amd import _lib, ops, runtime  # noqa: E402
from whisper_ipa_amd.runtime import ptr, sptr, use_stream  # noqa: E402

L = _lib.lib()
B, H, Tk = 64, 12, 1500
d = H * 64
g = torch.Generator(device="cuda").manual_seed(0)
xas = [torch.randn(B, Tk, d, device="cuda", generator=g).bfloat16() for _ in range(4)]
q = (torch.randn(B, d, device="cuda", generator=g) * 0.3).bfloat16()
wkT = (torch.randn(d, d, device="cuda", generator=g) * 0.05).bfloat16()
wv = (torch.randn(d, d, device="cuda", generator=g) * 0.05).bfloat16()
bv = torch.zeros(d, device="cuda")
out = torch.empty(B, d, device="cuda", dtype=torch.bfloat16)
nbytes = L.wipa_cross_absorbed_scratch_bytes(B, d, Tk)
scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
_lib.check(L.wipa_cross_absorbed_init(d))
_lib.check(L.wipa_cross_absorbed_attention(ptr(q), d, ptr(wkT), ptr(xas[0]), ptr(wv), ptr(bv), ptr(out), d, ptr(scratch), nbytes, B, H, d, Tk,
                                           64 ** -0.25, 0, sptr(torch.cuda.current_stream())))
torch.cuda.synchronize()
xs = torch.randn(64, 768, device="cuda", generator=g).bfloat16()
Ws = (torch.randn(768, 768, device="cuda", generator=g) * 0.05).bfloat16()
ys = torch.zeros(64, 768, device="cuda", dtype=torch.bfloat16)
xf = torch.randn(64, 768, device="cuda", generator=g)
lw, lb = torch.ones(768, device="cuda"), torch.zeros(768, device="cuda")
kv = (torch.randn(4, B, 2 * H, Tk, 64, device="cuda", generator=g) * 0.5).to(torch.bfloat16)


def stream_abs(n):
    s = sptr(runtime.stream())
    for i in range(n):
        _lib.check(L.wipa_cross_absorbed_stream(ptr(xas[i % 4]), ptr(scratch), nbytes, B, H, d, Tk, 0, s))


def cross_cached(n):
    for i in range(n):
        ops.decode_cross_attn(q, kv[i % 4])


def skinny(n):
    for _ in range(n):
        ops.gemm(xs, Ws, ys, M=64, N=768, K=768, lda=768, ldw=768, ldc=768)


def layernorm(n):
    for _ in range(n):
        ops.layernorm(xf, lw, lb, out_dtype=torch.bfloat16)


def timed(fa, na, fb=None, nb=0):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if fb is not None:
        with use_stream(2):
            fb(nb)
    with use_stream(1):
        fa(na)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for name, fa, na, fb, nb in (("skinny x absorbed stream", skinny, 3000, stream_abs, 600), ("layernorm x absorbed stream", layernorm, 3000, stream_abs, 600),
                             ("skinny x cached cross-attn", skinny, 3000, cross_cached, 400), ("absorbed x absorbed", stream_abs, 600, stream_abs, 600)):
    timed(fa, 8, fb, 2)
    ta, tb, tab = timed(fa, na), timed(fb, nb), timed(fa, na, fb, nb)
    print(f"{name:28s} A alone {ta:7.2f} ms  B alone {tb:7.2f} ms  together {tab:7.2f} ms  (serial sum {ta + tb:7.2f}, max {max(ta, tb):7.2f})", flush=True)
