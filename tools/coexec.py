#!/usr/bin/env python3
"""Do two kernels of DIFFERENT HIP streams make progress together?  Times N launches of kernel A alone, M launches of
kernel B alone, and both loops at once on two library streams (run on the GPU box; GPU_MAX_HW_QUEUES as in bench.py).
Pairs: decode cross-attention (HBM-bound) x encoder mlp1 GEMM (MFMA-bound); skinny projection x mlp1 GEMM; and the
cross-attention against itself."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_ipa_amd import ops  # noqa: E402
from whisper_ipa_amd import runtime  # noqa: E402
from whisper_ipa_amd.runtime import use_stream  # noqa: E402



def masked_stream(bits_of_8: int) -> torch.cuda.Stream:
    """A HIP stream restricted to the CUs whose index mod 8 is set in ``bits_of_8`` (hipExtStreamCreateWithCUMask)."""
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")
    word = sum(((bits_of_8 >> (i % 8)) & 1) << i for i in range(32))
    mask = (C.c_uint32 * 8)(*([word] * 8))
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


if len(sys.argv) > 2 and sys.argv[1] == "mask":  # coexec.py mask <8-bit pattern of stream 1 in hex>; stream 2 gets the complement
    dev = torch.cuda.current_device()
    torch.cuda.init()
    torch.zeros(1, device="cuda")
    pat = int(sys.argv[2], 16)
    runtime._streams[(dev, 1)] = masked_stream(pat)
    runtime._streams[(dev, 2)] = masked_stream(~pat & 0xFF)
    print(f"stream 1 on {bin(pat).count('1')}/8 of the CUs, stream 2 on the rest")
elif len(sys.argv) > 2:  # coexec.py <priority of stream 1> <priority of stream 2>   (lower = more urgent)
    dev = torch.cuda.current_device()
    runtime._streams[(dev, 1)] = torch.cuda.Stream(priority=int(sys.argv[1]))
    runtime._streams[(dev, 2)] = torch.cuda.Stream(priority=int(sys.argv[2]))

g = torch.Generator(device="cuda").manual_seed(0)
B, H, T, L = 64, 12, 1500, 12
kv = (torch.randn(L, B, 2 * H, T, 64, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
q = (torch.randn(B, H * 64, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
M, N, K = 96000, 3072, 768
A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
bias = torch.randn(N, device="cuda", generator=g)
C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
xs = torch.randn(64, 768, device="cuda", generator=g).bfloat16()
Ws = (torch.randn(768, 768, device="cuda", generator=g) * 0.05).bfloat16()
ys = torch.zeros(64, 768, device="cuda", dtype=torch.bfloat16)


def cross(n):
    for i in range(n):
        ops.decode_cross_attn(q, kv[i % L])


def gemm(n):
    for _ in range(n):
        ops.gemm(A, W, C, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, bias=bias, act=1)


def skinny(n):
    for _ in range(n):
        ops.gemm(xs, Ws, ys, M=64, N=768, K=768, lda=768, ldw=768, ldc=768)


def timed(fa, na, fb=None, nb=0):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if fb is not None:  # B first: its few long launches are queued before the host starts on A's many short ones
        with use_stream(2):
            fb(nb)
    with use_stream(1):
        fa(na)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for name, fa, na, fb, nb in (("cross x gemm", cross, 240, gemm, 20), ("skinny x gemm", skinny, 1500, gemm, 20),
                             ("cross x cross", cross, 240, cross, 240), ("skinny x cross", skinny, 1500, cross, 240)):
    timed(fa, 8, fb, 2)
    ta, tb, tab = timed(fa, na), timed(fb, nb), timed(fa, na, fb, nb)
    print(f"{name:15s} A alone {ta:7.2f} ms  B alone {tb:7.2f} ms  together {tab:7.2f} ms  (serial sum {ta + tb:7.2f}, max {max(ta, tb):7.2f})", flush=True)
