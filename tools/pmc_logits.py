#!/usr/bin/env python3
"""PMC target: the logits projection of a decode step, M = 64 rows, N = 51 865, K = 768, bf16 -> f32, ten plain launches.
Collect in SEPARATE passes of at most ~4 counters (one oversized --pmc set aborted rocprofv3 in round 1 with "Request exceeds
the capabilities of the hardware"), program directly after `--`:
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/f -- python3 tools/pmc_logits.py
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/w -- python3 tools/pmc_logits.py
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d out/s -- python3 tools/pmc_logits.py
(The library enqueues decode steps eagerly while counters are attached -- runtime.hip counters_attached() -- so whole-model
targets are safe too; this one only needs the GEMM.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_ipa_amd import ops  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
M, N, K = 64, 51865, 768
A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
out = torch.zeros(M, 51872, device="cuda")
GREEDY = os.environ.get("GREEDY", "0")  # 1: wipa_logits_greedy without logit stores (the lean step's launch); 2: with them
if GREEDY != "0":
    from whisper_ipa_amd import _lib
    from whisper_ipa_amd.runtime import ptr, sptr, stream as lib_stream

    L = _lib.lib()
    mask = torch.zeros(51872, device="cuda")
    pos = torch.tensor([5], dtype=torch.int32, device="cuda")
    nb = L.wipa_logits_greedy_partials_bytes(M)
    part = torch.empty(nb, dtype=torch.uint8, device="cuda")

    def launch(w):
        _lib.check(L.wipa_logits_greedy(ptr(A), K, ptr(w), K, ptr(out) if GREEDY == "2" else None, 51872, M, N, K, ptr(mask), ptr(mask), ptr(pos), 4,
                                        ptr(part), nb, sptr(lib_stream())), "wipa_logits_greedy")
else:
    def launch(w):
        ops.gemm(A, w, out, M=M, N=N, K=K, lda=K, ldw=K, ldc=51872)
torch.cuda.synchronize()  # the operands were made on torch's stream; the launches go to the library stream
for _ in range(10):
    launch(W)
torch.cuda.synchronize()
if "ROCPROF_COUNTER_COLLECTION" not in os.environ:  # plain run: event timing over graph-replayed launches
    from whisper_ipa_amd.runtime import stream

    s = stream()
    Ws = [W] + [W.clone() for _ in range(3)]  # cycle 4 copies (320 MB) so the weights are not cache resident
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        for w in Ws:
            launch(w)
    s.synchronize()
    with torch.cuda.graph(graph, stream=s):
        for i in range(40):
            launch(Ws[i % 4])
    with torch.cuda.stream(s):
        graph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        graph.replay()
        e1.record(s)
    e1.synchronize()
    us = e0.elapsed_time(e1) / 40 * 1e3
    bytes_alg = N * K * 2 + (M * 51872 * 4 if GREEDY != "1" else 0) + (M * 3 * 2048 * 4 if GREEDY != "0" else 0)
    print(f"logits GEMM GREEDY={GREEDY}: {us:.2f} us  {bytes_alg / 1e6:.1f} MB algorithmic  {bytes_alg / us / 1e6:.2f} TB/s")
print("done")
