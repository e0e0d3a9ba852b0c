#!/usr/bin/env python3
"""Decode groups x groups in flight on the headline workload (whisper-small bf16, 64-clip batches, 64 new tokens, fixed length):
ms per 64-clip batch through whisper_ipa_amd.pipeline.TranscribePipeline.  usage: python tools/group_sweep.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import whisper_ipa_amd  # noqa: E402,F401  (asks for 8 hardware queues before the GPU is touched)
import torch  # noqa: E402

import bench  # noqa: E402
from whisper_ipa_amd.pipeline import TranscribePipeline  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
model = bench.build_model("small")
audio = torch.from_numpy(bench.synthetic_audio(0, 64)).cuda()
opts = bench.bench_options()
ref = None
for P, G in ((4, 1), (2, 2), (3, 2), (4, 2), (2, 3), (2, 4), (1, 4), (3, 1), (4, 1)):
    n = steps // G * G
    with TranscribePipeline(model, opts, P, max_new_tokens=64, stop_on_eot=False, decode_group=G) as pipe:
        for _ in range(P * G):
            pipe.submit(audio)
        toks = [r.tokens for r in pipe.drain()]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            toks += [r.tokens for r in pipe.submit(audio)]
        toks += [r.tokens for r in pipe.drain()]
        dt = time.perf_counter() - t0
        splits = model.cross_splits
    same = all((t == toks[0]).all() for t in toks)
    if ref is None:
        ref = toks[0]
    print(f"groups in flight {P} x {G} batches per group ({P * G * 64:4d} clips in flight, {splits or 4} frame splits): "
          f"{1e3 * dt / n:7.2f} ms per 64-clip batch = {64 * 30.0 * n / dt:8.0f} audio-s/s | ids identical across batches {same}, "
          f"equal to the ungrouped run's {bool((toks[0] == ref).all())}", flush=True)
