#!/usr/bin/env python3
"""Throughput of the decoder fine-tune step (SURVEY config 3, one rank's share): whisper-small, fp32 master weights,
32 clips x 30 s per step, 64 target tokens, encoder frozen (forward only), masked CE, per-tensor clip, AdamW.
usage: python tools/train_bench.py [steps] [batch] [tokens]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_ipa_amd.training import DecoderTrainer  # noqa: E402
from whisper_ipa_amd.whisper import Whisper  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
T = int(sys.argv[3]) if len(sys.argv) > 3 else 64
import bench  # noqa: E402  (repo root: the synthetic whisper-small generator)

dims, W = bench.synthetic_weights_small(0)
g = torch.Generator().manual_seed(0)
m = Whisper(dims, dtype=torch.float32)
m.load_weights(W)
del W
tr = DecoderTrainer(m, lr=1e-5)
mel = (torch.randn(B, 3000, 80, generator=g) * 0.5).cuda()
EOT = 50257
tok = torch.randint(0, 50000, (B, T + 1), generator=g)
tok[:, 0] = 50258
tok[:, -3:] = EOT
tok = tok.cuda()


def sync():
    torch.cuda.synchronize()


for phase in ("encoder", "loss_and_grads", "update", "train_step"):
    ts = []
    for i in range(steps + 1):
        sync()
        t0 = time.perf_counter()
        if phase == "encoder":
            feats = m.embed_audio(mel)
        elif phase == "loss_and_grads":
            tr.loss_and_grads(feats, tok, EOT)
        elif phase == "update":
            tr.apply_update()
        else:
            loss = tr.train_step(mel, tok, EOT)
        sync()
        ts.append(time.perf_counter() - t0)
    ms = sorted(ts[1:])[len(ts[1:]) // 2] * 1e3
    print(f"{phase:16s} {ms:8.2f} ms   ({B / ms * 1e3:7.1f} clips/s)", flush=True)
print("loss", float(loss[0] if isinstance(loss, tuple) else loss))
