#!/usr/bin/env python3
"""BASELINE configs[3] / [4] through pipeline.TranscribePipeline at 1-4 passes in flight: ms per batch.
usage: python tools/other_configs_sweep.py"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import whisper_ipa_amd  # noqa: E402,F401
import torch  # noqa: E402

import bench  # noqa: E402
from whisper_ipa_amd.pipeline import TranscribePipeline  # noqa: E402

for name, B, weights, acts in (("medium", 256, "bf16", "bf16"), ("large-v3", 128, "fp8", "fp8")):
    model = bench.build_model(name, "bf16", weights, acts)
    audio = torch.from_numpy(bench.synthetic_audio(0, B)).cuda()
    for P in (1, 2, 3, 4):
        for splits in ((None,) if P == 1 else (None, 0)):  # None = the pipeline's rule (2 with >= 2 passes in flight), 0 = library default 4
            steps = 2 * P if P > 1 else 2
            with TranscribePipeline(model, bench.bench_options(), P, max_new_tokens=64, stop_on_eot=False, cross_splits=splits) as pipe:
                for _ in range(P):
                    pipe.submit(audio)
                for _ in pipe.drain():
                    pass
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    pipe.submit(audio)
                for _ in pipe.drain():
                    pass
                dt = time.perf_counter() - t0
                absorbed = bench.bench_absorbed(model, B)
            print(f"whisper-{name:8s} B={B:3d} weights {weights} | passes in flight {P} cross_splits {'rule' if splits is None else 4} "
                  f"({'absorbed' if absorbed else 'cached'}): {1e3 * dt / steps:8.2f} ms per batch = {B * 30.0 * steps / dt:7.0f} audio-s/s | "
                  f"HBM in use {torch.cuda.memory_allocated() / 2**30:5.1f} GiB", flush=True)
    del model, audio, pipe
    gc.collect()
    torch.cuda.empty_cache()
