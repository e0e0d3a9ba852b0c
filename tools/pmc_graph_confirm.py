#!/usr/bin/env python3
"""ONE bounded confirmation run for the round-1 hang (a captured decode step replayed while rocprofv3 --pmc is attached): a tiny
model (whisper-tiny, random weights, 2 clips) decodes 12 tokens with WIPA_DECODE_GRAPH=force, i.e. the step graph IS captured and replayed under the counters, and the ids
are compared with the eager path of the same process.  Run it once, under a timeout, never in a loop:

  WIPA_DECODE_GRAPH=force timeout -k 10 120 rocprofv3 --pmc SQ_WAVES --output-format csv -d out -- python3 tools/pmc_graph_confirm.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_ipa_amd.decoding import greedy_decode_tokens  # noqa: E402
from whisper_ipa_amd.whisper import Whisper  # noqa: E402

print("counters attached:", os.environ.get("ROCPROF_COUNTER_COLLECTION"), " WIPA_DECODE_GRAPH =", os.environ.get("WIPA_DECODE_GRAPH"), flush=True)
import bench  # noqa: E402  (its seeded weight generator; whisper-tiny: 4 + 4 layers, d = 384)

dims, W = bench.synthetic_weights_small(0, "tiny")
m = Whisper(dims, dtype=torch.bfloat16)
m.load_weights(W)
g = torch.Generator().manual_seed(0)
feats = (torch.randn(2, dims.n_audio_ctx, dims.n_audio_state, generator=g) * 0.5).to(torch.bfloat16).cuda()
init = [50258, 50259, 50359, 50363]
res_graph = greedy_decode_tokens(m, feats, init, [], [], 50257, max_new_tokens=12, stop_on_eot=False)
torch.cuda.synchronize()
print("graph path done:", res_graph.tokens[0, 4:].tolist(), flush=True)
os.environ["WIPA_DECODE_GRAPH"] = "0"
res_eager = greedy_decode_tokens(m, feats, init, [], [], 50257, max_new_tokens=12, stop_on_eot=False)
torch.cuda.synchronize()
print("eager path done:", res_eager.tokens[0, 4:].tolist(), flush=True)
print("ids equal:", bool((res_graph.tokens == res_eager.tokens).all()), flush=True)
