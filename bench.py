#!/usr/bin/env python3
"""bench.py -- whisper-small batched IPA transcription throughput on MI355X.

Metric (BASELINE.json): audio-seconds/sec transcribed, whisper-small, 30 s clips.
Workload at every N: configs[1] = "whisper-small bf16 batched inference, batch=64 x 30 s
synthetic clips" per GPU.  One "step" = one pass of the hot path over one batch already
resident in HBM: log-mel -> encoder -> cross-K/V -> prompt + 64 greedy decode positions
(KV cached, EOT latch on, no early stop so the work is fixed; the 4 prompt positions run as one batched
prefill pass, as the reference's decoder does) -> token ids on the host.
N > 1: clips are sharded data-parallel, one process per GPU, no data-path collective
(weak scaling); the only collective is the barrier / max-reduce of the timing itself.
THE TIMED REGION IS THE PRODUCT PATH: whisper_ipa_amd.pipeline.TranscribePipeline, the scheduler behind
`transcribe_batches(model, batches, options, passes_in_flight=...)` that scripts/evaluate_model.py and validate() call.
Up to `--pipeline` (default 4) passes are in flight on separate HIP streams with separate workspaces / decode states, each
doing ALL of its work inside the timed region, so the encoder of one batch overlaps the decode loop of the previous one
(`ms_per_step` = timed wall time / steps, i.e. the steady-state time per 64-clip batch).  The schedule -- not this file --
sets the streaming launch's frame splits (2 with several passes in flight) and the package asks for 8 hardware queues at import.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment makes this process a LAUNCHER: it starts N rank processes
(one per GPU, RCCL rendezvous on 127.0.0.1) without touching the GPU itself and relays rank 0's JSON line.  Under
`torch.distributed.run` (WORLD_SIZE set) it is a rank and WORLD_SIZE must equal --gpus.

Prints ONE JSON line (rank 0) with
  `roofline`       dominant HBM-bound kernel (decode-step cross-attention), HIP-event timed on its launch stream;
  `roofline_mfma`  the encoder + cross-K/V GEMM set (19.4 TFLOP algorithmic per 64-clip pass) against the dense bf16 MFMA
                   peak, from HIP events around every GEMM launch of a real pass (wipa_profile_begin/end);
  `decode_step`    one whole decode step (graph replay, ONE pass in flight) against the HBM peak: SURVEY section 8d bytes
                   (B x cross-K/V + self-K/V + decoder weights) / event-timed step;
  `evaluate_style` the same workload the way the reference's batch caller consumes it (transcribe_batches with early stop armed, every
                   row turned into text), as a fraction of `value`;
  `parity_vs_cpu`  token ids of clips 0..7 against the CPU oracle's ids for the same weights and clips (+ `parity_vs_cpu_peaky`);
  `cpu_baseline`   the CPU oracle (a port) timed on this box's host cores on a bounded sample (rank 0, N = 1 only);
  `finetune_step`  the decoder fine-tune step (exact f32 products), `other_configs` BASELINE configs[3] / [4] through the same path.
`--mode train` times the decoder fine-tune step instead (one rank's share of BASELINE.json configs[2]).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The passes in flight live on several HIP streams; ROCm multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4)
# and reads the variable when the HIP runtime starts.  Importing the package asks for 8 (runtime.request_hw_queues: measured
# with 4 passes in flight, 4 queues 86.5-88 ms per pass, 8 queues 72 ms) -- nothing in this file sets it; `config.hw_queues`
# in the line is what the package reports.  Imported before anything can touch the GPU.
import whisper_ipa_amd  # noqa: E402,F401

T_START = time.time()


def log(msg: str) -> None:
    """progress to stderr (the JSON line is the only thing on stdout)"""
    print(f"[bench {time.time() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def host_cores() -> int:
    """cores this process may really use: affinity mask, cgroup quota, capped at the GPU box's
    16-core share (os.cpu_count() reports the whole host and oversubscribes torch)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def bench_absorbed(model, B: int) -> bool:
    """the decode-step cross-attention form this run's (batch, --new-tokens) takes: cross_attention='auto' decides by the
    measured table (Whisper.use_absorbed), exactly as greedy_launch does for the timed passes"""
    return model.use_absorbed(B, NEW_TOKENS)


def bench_packed(model, B: int):
    return model.packed(absorbed=bench_absorbed(model, B))


def bench_state(model, B: int):
    from whisper_ipa_amd.decoding import _state_for

    return _state_for(model, B, bench_packed(model, B))


def rank_threads() -> int:
    """torch CPU threads of THIS rank: the host's usable cores shared among the ranks on it (parallel.host_threads_per_rank)"""
    from whisper_ipa_amd.parallel import host_threads_per_rank

    return host_threads_per_rank()


BATCH = 64          # clips per GPU
N_PIPELINE = 4      # consecutive passes kept in flight (pipeline.transcribe_batches(passes_in_flight=...)).  Measured r01 with
                    # 8 hardware queues (ms per 64-clip pass, repeatable to 0.5 %): 1 -> 133.2, 3 -> 98.8, 4 -> 94.7,
                    # 5 -> 123.5: the MFMA-bound encoder of later passes runs in the shadows of the launch/HBM-bound
                    # decode loops of earlier ones
NEW_TOKENS = 64     # decode positions per clip (SURVEY.md section 8d primary setting)
HBM_PEAK_GBS = 8000.0
MFMA_BF16_PEAK_TFLOPS = 2500.0
MFMA_FP8_PEAK_TFLOPS = 5000.0   # dense fp8 MFMA (MI355X_MICROARCH.md); the headline figures with 2:1 sparsity are never used


def model_dims(name: str = "small"):
    """Published Whisper sizes.  "small" is the benchmark configuration; the others are for sizing runs
    (`--model medium --batch 256` is BASELINE.json configs[3])."""
    from whisper_ipa_amd.whisper import ModelDimensions

    table = {"tiny": (80, 384, 6, 4, 51865), "base": (80, 512, 8, 6, 51865), "small": (80, 768, 12, 12, 51865),
             "medium": (80, 1024, 16, 24, 51865), "large-v3": (128, 1280, 20, 32, 51866)}
    n_mels, d, heads, layers, vocab = table[name]
    return ModelDimensions(n_mels, 1500, d, heads, layers, vocab, 448, d, heads, layers)


def small_dims():
    return model_dims("small")


PEAKY_GAIN = 128.0


def peaky_positional_table(W, dims, seed: int, suppress_always, gain: float = PEAKY_GAIN):
    """decoder.positional_embedding of the "peaky" preset: the lively table + gain * token_embedding[pi(p)], pi a seeded draw of
    distinct never-suppressed text tokens -- the token after position p is pi(p) with a top-1 margin of several logit standard
    deviations (a confident model; the ids are decided by position, the audio still moves every logit).  Same recipe as the
    oracle's generator (tests/test_host_logic.py checks the equality)."""
    g = torch.Generator().manual_seed(seed + 12345)
    banned = set(int(t) for t in suppress_always)
    cand = torch.tensor([i for i in range(1000, 50000) if i not in banned])
    pi = cand[torch.randperm(len(cand), generator=g)[: dims.n_text_ctx]]
    return W["decoder.positional_embedding"] + gain * W["decoder.token_embedding.weight"][pi].float()


def synthetic_weights_small(seed: int = 0, name: str = "small", preset: str = "lively"):
    """Random-init whisper-<name> in mlx_whisper naming (no checkpoint exists offline).  Same
    recipe as the oracle's generator, restated here so the product path does not import it.
    preset "peaky": see peaky_positional_table."""
    dims = model_dims(name)
    g = torch.Generator().manual_seed(seed)
    std, emb_std, pos_std, out_scale = 0.06, 0.2, 1.2, 4.0

    def rn(*shape, s=std):
        return torch.randn(*shape, generator=g) * s

    W = {}
    d = dims.n_audio_state
    W["encoder.conv1.weight"] = rn(d, 3, dims.n_mels, s=0.05)
    W["encoder.conv1.bias"] = rn(d)
    W["encoder.conv2.weight"] = rn(d, 3, d)
    W["encoder.conv2.bias"] = rn(d)

    def block(prefix, d, cross):
        for a in ["attn"] + (["cross_attn"] if cross else []):
            W[f"{prefix}.{a}.query.weight"] = rn(d, d)
            W[f"{prefix}.{a}.query.bias"] = rn(d)
            W[f"{prefix}.{a}.key.weight"] = rn(d, d)
            W[f"{prefix}.{a}.value.weight"] = rn(d, d)
            W[f"{prefix}.{a}.value.bias"] = rn(d)
            W[f"{prefix}.{a}.out.weight"] = rn(d, d)
            W[f"{prefix}.{a}.out.bias"] = rn(d)
            W[f"{prefix}.{a}_ln.weight"] = 1.0 + rn(d, s=0.1)
            W[f"{prefix}.{a}_ln.bias"] = rn(d, s=0.1)
        W[f"{prefix}.mlp1.weight"] = rn(4 * d, d)
        W[f"{prefix}.mlp1.bias"] = rn(4 * d)
        W[f"{prefix}.mlp2.weight"] = rn(d, 4 * d)
        W[f"{prefix}.mlp2.bias"] = rn(d)
        W[f"{prefix}.mlp_ln.weight"] = 1.0 + rn(d, s=0.1)
        W[f"{prefix}.mlp_ln.bias"] = rn(d, s=0.1)

    for i in range(dims.n_audio_layer):
        block(f"encoder.blocks.{i}", d, False)
    W["encoder.ln_post.weight"] = 1.0 + rn(d, s=0.1)
    W["encoder.ln_post.bias"] = rn(d, s=0.1)
    W["decoder.token_embedding.weight"] = rn(dims.n_vocab, d, s=emb_std)
    W["decoder.positional_embedding"] = rn(dims.n_text_ctx, d, s=pos_std)
    for i in range(dims.n_text_layer):
        block(f"decoder.blocks.{i}", d, True)
    for k in list(W):
        if k.startswith("decoder.blocks.") and k.split(".")[-2] in ("out", "mlp2"):
            W[k] = W[k] * out_scale
    W["decoder.ln.weight"] = 1.0 + rn(d, s=0.1)
    W["decoder.ln.bias"] = rn(d, s=0.1)
    if preset == "peaky":
        W["decoder.positional_embedding"] = peaky_positional_table(W, dims, seed, decode_setup(dims.n_vocab - 51765 - 1)[1])
    else:
        assert preset == "lively", preset
    return dims, W


def synthetic_audio(first_clip: int, n: int) -> np.ndarray:
    """default_rng(1234 + clip), 0.1 * N(0,1), 480 000 samples, stored as f32 -- sample for sample
    the clips the CPU oracle generates (oracle.whisper_ref.synthetic_clip; tests/test_host_logic.py checks the equality),
    so `parity_vs_cpu` compares like with like."""
    out = np.empty((n, 480000), dtype=np.float32)
    for i in range(n):
        out[i] = (0.1 * np.random.default_rng(1234 + first_clip + i).standard_normal(480000)).astype(np.float32)
    return out


# multilingual special ids / suppress list (tokenizer constants, whisper_ipa_amd/tokenizer.py)
def decode_setup(num_languages: int = 99):
    from whisper_ipa_amd.tokenizer import get_tokenizer
    from whisper_ipa_amd.decoding import DecodingOptions, _suppress_lists

    tok = get_tokenizer(True, num_languages=num_languages)
    always, first = _suppress_lists(DecodingOptions(language="en", without_timestamps=True), tok)
    return list(tok.sot_sequence_including_notimestamps), always, first, tok.eot


def bench_options():
    """what the reference's inference callers pass (scripts/transcribe_single.py:49-52, scripts/evaluate_model.py:170-173)"""
    from whisper_ipa_amd.decoding import DecodingOptions

    return DecodingOptions(language="en", without_timestamps=True)


def one_pass(model, audio) -> np.ndarray:
    """ONE batch through the product path with nothing else in flight (pipeline.transcribe_batches(passes_in_flight=1): log-mel
    -> encoder -> prompt + NEW_TOKENS greedy steps, the model's own cross_splits); returns the token matrix on the host."""
    from whisper_ipa_amd.pipeline import transcribe_batches

    (r,) = list(transcribe_batches(model, [audio], bench_options(), passes_in_flight=1, max_new_tokens=NEW_TOKENS, stop_on_eot=False))
    return r.tokens


def phase_enc_loop(model, audio, steps: int, pipeline: int) -> None:
    """DIAGNOSTIC (--phase enc): log-mel + encoder alone on the pipeline's stream sets, `pipeline` of them in flight"""
    from whisper_ipa_amd import audio as A
    from whisper_ipa_amd.runtime import use_stream

    pending = {}
    for i in range(steps):
        slot = i % pipeline
        if slot in pending:
            pending.pop(slot).synchronize()
        with use_stream(slot) as s:
            model.encode_padded(A.log_mel_padded(audio, model.dims.n_mels, model.dtype), audio.shape[0])
            ev = torch.cuda.Event()
            ev.record(s)
            pending[slot] = ev
    for ev in pending.values():
        ev.synchronize()


def latest_profile(suffix: str, fallback: str = ""):
    """path of the newest committed counter summary profiles/r<NN>_<suffix> (counters are collected once per round by
    tools/profile.sh + tools/summaries.py, in separate rocprofv3 --pmc passes; bench.py only reads them)"""
    import glob

    found = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{suffix}")))
    return found[-1] if found else os.path.join(ROOT, "profiles", fallback)


def roofline_cross_attn(model, B: int, iters: int = 48):
    """Dominant HBM-bound kernel: decode-step cross-attention (K11).  Algorithmic bytes per launch
    = B * 2 * H * 1500 * 64 * sizeof(bf16) (every cached K and V element once) + q/out.
    Timed with events on the library stream over back-to-back launches that cycle through all
    layer caches (each launch streams bytes no other recent launch touched -> HBM, not cache)."""
    from whisper_ipa_amd import ops
    from whisper_ipa_amd.runtime import on_stream, stream

    d = model.dims
    H, Ta = d.n_text_head, d.n_audio_ctx
    st = bench_state(model, B)
    lay = st.layout
    if bench_absorbed(model, B):
        return roofline_cross_absorbed(model, B, st, iters)
    e = 2 if model.dtype == torch.bfloat16 else 4
    per_layer = B * 2 * H * Ta * 64
    kv_all = st.blob[lay.cross_kv: lay.cross_kv + d.n_text_layer * per_layer * e].view(model.dtype).view(
        d.n_text_layer, B, 2 * H, Ta, 64)
    from whisper_ipa_amd import _lib
    from whisper_ipa_amd.runtime import dt_code, ptr, sptr

    L = _lib.lib()
    with on_stream():
        q = torch.randn(B, H * 64, device=model.device).to(model.dtype)
        out = torch.empty_like(q)
        for l in range(d.n_text_layer):
            ops.decode_cross_attn(q, kv_all[l])
        s = stream()
        s.synchronize()
        # the launches are captured into a graph so the event pair times the kernels, not the
        # Python/ctypes launch path (which costs more than the 50 us kernel)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            for i in range(iters):
                _lib.check(L.wipa_decode_cross_attn(ptr(q), ptr(kv_all[i % d.n_text_layer]), ptr(out), B, H, Ta,
                                                    dt_code(model.dtype), sptr(s)))
        graph.replay()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(s)
        graph.replay()
        ev1.record(s)
        ev1.synchronize()
    ms = ev0.elapsed_time(ev1) / iters
    bytes_alg = per_layer * e + 2 * B * H * 64 * e
    achieved = bytes_alg / (ms * 1e-3) / 1e9
    streaming = {"kernel": "decode_attn_kernel (the streaming loop alone: cross-attention with a given query)",
                 "avg_launch_ms": round(ms, 5), "achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4)}
    fused = os.environ.get("WIPA_DECODE_FUSED", "2") != "0" and model.weights_format != "fp8_e4m3"
    if fused:
        # the kernel the default decode step actually launches: slab sum + residual + LayerNorm + cross query + the same
        # streaming loop (wipa_decode_cross_block).  Same K/V bytes; the rows and the query weights add 0.6 %.
        import ctypes as C

        dd = d.n_text_state
        with on_stream():
            s = stream()
            x = torch.randn(B, dd, device=model.device)
            x_out = torch.empty_like(x)
            slabs = torch.randn(2, B, dd, device=model.device) * 0.3
            pk = bench_packed(model, B)
            desc = []
            for l in range(d.n_text_layer):
                lw = pk["dec"][_lib.DEC_GLOBAL + _lib.DEC_PER_LAYER * l: _lib.DEC_GLOBAL + _lib.DEC_PER_LAYER * (l + 1)]
                c = _lib.CrossBlockDesc()
                c.x_in, c.x_out, c.slabs, c.bias_o, c.ln_w, c.ln_b = ptr(x), ptr(x_out), ptr(slabs), None, ptr(lw[6]), ptr(lw[7])
                c.wq, c.bq, c.kv, c.out = ptr(lw[8]), ptr(lw[9]), ptr(kv_all[l]), ptr(out)
                c.slab_stride = B * dd
                c.n_slabs, c.B, c.d, c.H, c.Tk, c.dtype, c.eps, c.qk_scale = 2, B, dd, H, Ta, dt_code(model.dtype), 1e-5, 64 ** -0.25
                desc.append(c)
            for c in desc:
                _lib.check(L.wipa_decode_cross_block(C.byref(c), sptr(s)))
            s.synchronize()
            graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph2, stream=s):
                for i in range(iters):
                    _lib.check(L.wipa_decode_cross_block(C.byref(desc[i % d.n_text_layer]), sptr(s)))
            graph2.replay()
            ev0.record(s)
            graph2.replay()
            ev1.record(s)
            ev1.synchronize()
        ms = ev0.elapsed_time(ev1) / iters
        bytes_alg = per_layer * e + B * dd * (4 + 2 * 4 + 4 + e) + dd * dd * e
        achieved = bytes_alg / (ms * 1e-3) / 1e9
        # ... and the three launches it replaces (round 1's step): slab sum + LayerNorm, cross-query GEMM, streaming loop
        with on_stream():
            ln_out = torch.empty(B, dd, device=model.device, dtype=model.dtype)
            graph3 = torch.cuda.CUDAGraph()

            def unfused(i):
                lw = pk["dec"][_lib.DEC_GLOBAL + _lib.DEC_PER_LAYER * (i % d.n_text_layer): _lib.DEC_GLOBAL + _lib.DEC_PER_LAYER * (i % d.n_text_layer + 1)]
                _lib.check(L.wipa_add_slabs_layernorm(ptr(x), dd, ptr(slabs), 2, B * dd, ptr(ln_out), dt_code(model.dtype), dd,
                                                      ptr(lw[6]), ptr(lw[7]), B, dd, 1e-5, sptr(s)))
                ops.gemm(ln_out, lw[8], q, M=B, N=dd, K=dd, lda=dd, ldw=dd, ldc=dd, bias=lw[9], col_scale_n=dd, col_scale=64 ** -0.25)
                _lib.check(L.wipa_decode_cross_attn(ptr(q), ptr(kv_all[i % d.n_text_layer]), ptr(out), B, H, Ta, dt_code(model.dtype), sptr(s)))

            for i in range(d.n_text_layer):
                unfused(i)
            s.synchronize()
            with torch.cuda.graph(graph3, stream=s):
                for i in range(iters):
                    unfused(i)
            graph3.replay()
            ev0.record(s)
            graph3.replay()
            ev1.record(s)
            ev1.synchronize()
        ms3 = ev0.elapsed_time(ev1) / iters
        replaced = {
            "kernels": "add_slabs_layernorm + cross-query gemm_skinny + decode_attn_kernel (the unfused step, WIPA_DECODE_FUSED=0)",
            "avg_ms": round(ms3, 5), "frac": round(bytes_alg / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    # HBM traffic per launch from the PMC counters (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE,
    # separate rocprofv3 --pmc passes over tools/pmc_cross_attn.py; summary committed under profiles/).  Only
    # quoted when it was collected for exactly this launch shape.
    traffic = None
    try:
        pm = json.load(open(latest_profile("pmc_cross_block.json") if fused else os.path.join(ROOT, "profiles", "r01_pmc_cross_attn.json")))
        if pm.get("algorithmic_bytes_per_launch") == bytes_alg:
            traffic = pm["hbm_bytes_per_launch"]
    except Exception:
        pass
    staged = (H * B <= 768 and d.n_text_state <= 768 and model.dtype == torch.bfloat16
              and os.environ.get("WIPA_CROSS_PRE", "4") in ("4", "6"))  # wipa_decode_cross_block's dispatch rule (decode_fused.hip)
    name = (("decode_cross_block_pre_kernel" if staged else "decode_cross_block_kernel") +
            " (decode-step cross-attention incl. slab sum + LayerNorm + cross query" + (", first K / V rows staged by LDS-DMA under the prologue)" if staged else ")") if fused
            else "decode_attn_kernel (decode-step cross-attention)")
    out = {"kernel": name, "bound": "hbm", "achieved": round(achieved, 1),
           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
           "algorithmic_bytes_per_launch": bytes_alg, "avg_launch_ms": round(ms, 5), "streaming_loop_alone": streaming}
    if fused:
        out["three_launches_it_replaces"] = replaced
    return out


def roofline_cross_absorbed(model, B: int, st, iters: int = 48):
    """Dominant HBM-bound kernel of the decode step with --cross-attention absorbed: cross_absorbed_v2_kernel, the ONE pass over the encoder
    output xa that serves scores and values of a layer (csrc/cross_absorbed.hip).  Algorithmic bytes per launch = B * 1500 * d *
    sizeof(bf16) (every element of xa once) + the absorbed queries and the split partials.  SURVEY.md section 8(d) counted
    55.3 MB per clip and step for the cached K and V of the 12 layers; with the projections absorbed the bytes a step MUST read
    are half of that, which is the point -- the line also gives the rate in cached-K/V terms (`kv_equivalent`).  Timed with
    events around graph-replayed launches in the decode loop's own order: the 12 layers of a step read the same encoder
    output, then the step of another in-flight pass reads its own (4 outputs in turn: 590 MB against 256 MB of Infinity Cache,
    so the first launch of every group is cold and the others largely cache-fed -- rotating a different output into every
    launch gives 34.6 us instead of 30.6, one output alone 30.3; the kernel trace of the real run averages 30.3)."""
    import ctypes as C

    from whisper_ipa_amd import _lib
    from whisper_ipa_amd.runtime import on_stream, ptr, sptr, stream

    L = _lib.lib()
    d = model.dims
    H, Ta, dd = d.n_text_head, d.n_audio_ctx, d.n_text_state
    pk = model.packed(absorbed=True)
    # launch order of the real decode loop: the n_layer launches of a step read the SAME encoder output (the first finds it cold,
    # the others largely in the 256 MB Infinity Cache), then another in-flight pass's step runs on its own encoder output
    n_buf, per_buf = N_PIPELINE, d.n_text_layer  # four encoder outputs in turn, whatever --pipeline says (comparable across runs)
    with on_stream() as s:
        xas = [torch.randn(B, Ta, dd, device=model.device).to(torch.bfloat16) for _ in range(n_buf)]
        q = (torch.randn(B, dd, device=model.device) * 0.3).to(torch.bfloat16)
        out = torch.empty_like(q)
        nbytes = L.wipa_cross_absorbed_scratch_bytes(B, dd, Ta)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=model.device)
        base = _lib.DEC_GLOBAL + _lib.DEC_PER_LAYER * d.n_text_layer
        lw = pk["dec"][_lib.DEC_GLOBAL: _lib.DEC_GLOBAL + _lib.DEC_PER_LAYER]
        wkT, wkv, bkv = pk["dec"][base], lw[10], lw[11]

        def full(i):
            _lib.check(L.wipa_cross_absorbed_attention(ptr(q), dd, ptr(wkT), ptr(xas[(i // per_buf) % n_buf]), ptr(wkv[dd:]), ptr(bkv[dd:]), ptr(out), dd,
                                                       ptr(scratch), nbytes, B, H, dd, Ta, 64 ** -0.25, model.cross_splits, sptr(s)), "wipa_cross_absorbed_attention")

        def stream_only(i):
            _lib.check(L.wipa_cross_absorbed_stream(ptr(xas[(i // per_buf) % n_buf]), ptr(scratch), nbytes, B, H, dd, Ta, model.cross_splits, sptr(s)),
                       "wipa_cross_absorbed_stream")

        times = {}
        for name, fn in (("stream", stream_only), ("layer_call", full)):
            for i in range(n_buf):
                full(i)
            s.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=s):
                for i in range(iters):
                    fn(i)
            graph.replay()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(s)
            graph.replay()
            ev1.record(s)
            ev1.synchronize()
            times[name] = ev0.elapsed_time(ev1) / iters
        S = L.wipa_cross_absorbed_splits(model.cross_splits, Ta)  # the streaming launch of the decode steps this run timed
        pair_ms = None
        if S * B <= 128:
            # a launch of <= 128 workgroups is MEANT to share the chip with another pass's launch: the same launches on two HIP
            # streams at once (separate scratch, the same encoder outputs), aggregate bytes over the wall time of both
            s2 = stream(1)  # library stream 1 = where the second pass in flight decodes (its own hardware queue, like in the timed run)
            scratch2 = scratch.clone()
            s.synchronize()
            graphs = []
            for st_, sc_ in ((s, scratch), (s2, scratch2)):
                g_ = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_, stream=st_):
                    for i in range(iters):  # the second stream two encoder outputs behind the first: nothing shared at any time
                        _lib.check(L.wipa_cross_absorbed_stream(ptr(xas[(i // per_buf + (2 if st_ is s2 else 0)) % n_buf]), ptr(sc_), nbytes, B, H, dd, Ta,
                                                                model.cross_splits, st_.cuda_stream), "wipa_cross_absorbed_stream")
                graphs.append((st_, g_))
            for rep in range(2):
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record(s)
                s2.wait_event(ev0)
                for st_, g_ in graphs:
                    with torch.cuda.stream(st_):
                        g_.replay()
                s.wait_stream(s2)
                ev1.record(s)
                ev1.synchronize()
            pair_ms = ev0.elapsed_time(ev1) / iters  # wall time per PAIR of launches
    xa_bytes = B * Ta * dd * 2
    bytes_alg = xa_bytes + B * 16 * dd * 2 + B * S * (H * dd + 32) * 4  # xa once + absorbed queries in + split partials (H head rows, m, l) out
    ms = times["stream"]
    achieved = bytes_alg / (ms * 1e-3) / 1e9
    traffic = None
    try:
        pm = json.load(open(latest_profile("pmc_cross_absorbed.json" if S == 4 else f"pmc_cross_absorbed_s{S}.json")))
        if pm.get("algorithmic_bytes_per_launch") == bytes_alg:
            traffic = pm["hbm_bytes_per_launch"]
    except Exception:
        pass
    kv_bytes = B * 2 * H * Ta * 64 * 2
    return {"kernel": "cross_absorbed_v2_kernel (decode-step cross-attention of one layer: one pass over the encoder output, key / value "
                      "projections absorbed into the query and the output)", "bound": "hbm", "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "algorithmic_bytes_per_launch": bytes_alg, "avg_launch_ms": round(ms, 5),
            "launch_order": f"{per_buf} consecutive launches (the layers of one decode step) per encoder output, {n_buf} encoder outputs (passes in flight) in turn",
            "frame_splits": S, "workgroups": S * B,
            **({"note": f"this launch holds {S * B} of the 256 CUs BY DESIGN (Whisper.cross_splits = {S}: the setting for several passes in flight): `frac` is one "
                        "launch alone against the whole chip's HBM peak; `two_launches_side_by_side` is the chip-level rate in the situation the setting "
                        "exists for, `library_default_setting` the same kernel launched on all 256 CUs"} if S * B <= 128 else {}),
            **({"two_launches_side_by_side": {
                "what": f"the launch holds {S * B} of the 256 CUs by design (half-chip launches for several passes in flight): the same launches on "
                        "two HIP streams at once, aggregate algorithmic bytes over the wall time",
                "avg_pair_ms": round(pair_ms, 5), "achieved": round(2 * bytes_alg / (pair_ms * 1e-3) / 1e9, 1),
                "frac": round(2 * bytes_alg / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}} if pair_ms else {}),
            "layer_call": {"what": "absorb-q + streaming + merge / value projection (3 launches: what replaces the cached-K/V cross "
                                   "block's streaming loop)", "avg_ms": round(times["layer_call"], 5)},
            "kv_equivalent": {"what": "the cached K / V bytes this launch stands for (SURVEY.md 8d: 55.3 MB per clip and step over 12 "
                                      "layers) over the same time", "bytes": kv_bytes,
                              "GB/s": round(kv_bytes / (ms * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(kv_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}


def roofline_mfma(model, audio, pmc_file: str = "pmc_encoder_gemm.json"):
    """The MFMA-bound kernel set: every GEMM / conv-as-GEMM of one encoder pass plus the cross-K/V projection.
    Algorithmic FLOPs per clip are SURVEY.md App. B's (whisper-small: 261.2 + 42.5 GFLOP); the time is the sum of HIP-event
    spans around each GEMM launch of a real pass on the library stream (wipa_profile_begin / wipa_profile_end)."""
    import ctypes as C

    from whisper_ipa_amd import _lib, audio as A
    from whisper_ipa_amd.decoding import _state_for
    from whisper_ipa_amd.runtime import on_stream, ptr, sptr

    d = model.dims
    B, Ta, de, dd = audio.shape[0], d.n_audio_ctx, d.n_audio_state, d.n_text_state
    # 2*M*N*K of: conv1 (K = 3 n_mels, 3000 frames), conv2 (K = 3 d, 1500 frames), per layer q,k,v,out (4 d^2) + MLP (8 d^2),
    # and the decoder's cross key / value projections of the encoder output (2 d^2 per decoder layer)
    enc = 2.0 * (3000 * de * 3 * d.n_mels + Ta * de * 3 * de + d.n_audio_layer * Ta * 12 * de * de)
    ckv = 0.0 if bench_absorbed(model, B) else 2.0 * d.n_text_layer * Ta * 2 * dd * dd  # absorbed projections: no cross-K/V GEMMs
    flops = B * (enc + ckv)
    L = _lib.lib()
    pk = bench_packed(model, B)
    ms = (C.c_float * 4)()
    cnt = (C.c_int * 4)()
    best = None
    for _ in range(3):
        with on_stream() as s:
            st = bench_state(model, B)
            _lib.check(L.wipa_profile_begin(sptr(s)), "wipa_profile_begin")
            try:
                mel = A.log_mel_padded(audio, d.n_mels, model.dtype)
                feats = model.encode_padded(mel, B)
                _lib.check(L.wipa_decoder_set_audio(C.byref(pk["cfg"]), pk["dec_tab"], ptr(feats), ptr(st.blob), st.blob.numel(), B, sptr(s)),
                           "wipa_decoder_set_audio")
            finally:
                _lib.check(L.wipa_profile_end(ms, cnt), "wipa_profile_end")
        if best is None or ms[0] < best[0]:
            best = (float(ms[0]), int(cnt[0]), float(ms[1]), float(ms[2]))
    gemm_ms, n_gemm, attn_ms, norm_ms = best
    achieved = flops / (gemm_ms * 1e-3) / 1e12
    attn_flops = B * d.n_audio_layer * 4.0 * Ta * Ta * de
    out = {"kernel": "gemm_nt384/gemm_nt256 (encoder GEMM + conv" + ("" if bench_absorbed(model, B) else " + cross-K/V projection") + " set)", "bound": "mfma",
           "achieved": round(achieved, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "algorithmic_tflop_per_pass": round(flops / 1e12, 3),
           "gemm_ms_per_pass": round(gemm_ms, 3), "gemm_launches": n_gemm,
           "flash_attention": {"ms_per_pass": round(attn_ms, 3), "achieved": round(attn_flops / (attn_ms * 1e-3) / 1e12, 1),
                               "frac": round(attn_flops / (attn_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)},
           "layernorm_ms_per_pass": round(norm_ms, 3), "mfma_busy_pmc": None}
    if model.activations_format == "fp8_e4m3":
        # 11/12 of the encoder's layer GEMM FLOPs (q|k, value, mlp1, mlp2) run fp8 x fp8; the out projection, the convolutions
        # and the cross-K/V projection stay bf16.  The set's roofline is the time both shares would take at THEIR dense peaks:
        # peak = FLOPs / (fp8 FLOPs / 5 PF/s + bf16 FLOPs / 2.5 PF/s) -- never the bf16 peak for work that runs on the fp8 MFMA.
        f8 = B * 2.0 * d.n_audio_layer * Ta * 11 * de * de
        eff_peak = flops / (f8 / MFMA_FP8_PEAK_TFLOPS + (flops - f8) / MFMA_BF16_PEAK_TFLOPS)
        out["peak"] = round(eff_peak, 1)
        out["frac"] = round(achieved / eff_peak, 4)
        out["fp8_mfma"] = {"fp8_share_of_flops": round(f8 / flops, 4), "peak_fp8": MFMA_FP8_PEAK_TFLOPS, "peak_bf16": MFMA_BF16_PEAK_TFLOPS,
                           "frac_of_fp8_peak": round(achieved / MFMA_FP8_PEAK_TFLOPS, 4), "frac_of_bf16_peak": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4),
                           "note": "whole GEMM set (fp8 and bf16 launches together) / its summed launch time; `peak` above is the FLOP-weighted blend "
                                   "of the two dense peaks, `frac` is against it"}
    if d.n_audio_state == 768 and model.activations_format != "fp8_e4m3" and model.dtype == torch.bfloat16:
        try:  # whisper-small bf16: SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE per GEMM shape, separate rocprofv3 --pmc passes (profiles/)
            out["mfma_busy_pmc"] = json.load(open(latest_profile(pmc_file)))
        except Exception:
            pass
    return out


def decode_step_roofline(model, B: int, n_steps: int = 48):
    """One WHOLE decode step (all layers + logits + greedy update: the captured hipGraph the decode loop replays) with one
    pass in flight, against the HBM peak.  Bytes per step as SURVEY.md section 8d counts them: B x cross-K/V (every cached
    key and value once) + B x self-K/V of the positions filled so far + the decoder weights once."""
    import ctypes as C

    from whisper_ipa_amd import _lib
    from whisper_ipa_amd.decoding import _mask, _state_for
    from whisper_ipa_amd.runtime import on_stream, ptr, sptr

    L = _lib.lib()
    pk = bench_packed(model, B)
    absorbed = bench_absorbed(model, B)
    d = model.dims
    e = 2 if model.dtype == torch.bfloat16 else 4
    init, always, first, eot = decode_setup()
    m_always, m_first = _mask(model, always), _mask(model, list(always) + list(first))
    with on_stream() as s:
        st = bench_state(model, B)  # holds the caches of the last pass on library stream 0
        p0 = int(st.pos.cpu())
        n_steps = max(1, min(n_steps, d.n_text_ctx - 2 - p0))
        args = (C.byref(pk["cfg"]), pk["dec_tab"], ptr(st.blob), st.blob.numel(), B, len(init), eot, ptr(m_first), ptr(m_always))
        _lib.check(L.wipa_decoder_run(*args, 2, 1, sptr(s)), "wipa_decoder_run")
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(s)
        _lib.check(L.wipa_decoder_run(*args, n_steps, 1, sptr(s)), "wipa_decoder_run")
        ev1.record(s)
        ev1.synchronize()
    ms = ev0.elapsed_time(ev1) / n_steps
    t_mean = p0 + 2 + n_steps / 2.0
    dd, Ld = d.n_text_state, d.n_text_layer
    cross = B * Ld * 2 * d.n_audio_ctx * dd * e
    if absorbed:  # one pass over the encoder output per layer instead of one over K and one over V
        cross = B * Ld * d.n_audio_ctx * dd * e
    self_kv = B * Ld * 2 * t_mean * dd * e
    dec_params = d.n_vocab * dd + d.n_text_ctx * dd + Ld * (4 * dd * dd + 4 * dd * dd + 8 * dd * dd) + Ld * 11 * dd + 2 * dd
    weights = dec_params * e
    if model.weights_format == "fp8_e4m3":  # what the step streams: 1-byte codes for everything but the cross K/V projection
        weights = d.n_vocab * dd + Ld * 14 * dd * dd + (d.n_text_ctx * dd + Ld * 11 * dd + 2 * dd) * 4
    total = cross + self_kv + weights
    achieved = total / (ms * 1e-3) / 1e9
    out = {"what": "one decode step, hipGraph replay, one pass in flight", "bound": "hbm", "ms_per_step": round(ms, 4),
           "bytes_per_step": int(total), "cross_kv_bytes": int(cross), "self_kv_bytes": int(self_kv), "weight_bytes": int(weights),
           "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
           "cross_attention": "absorbed projections: the encoder output streamed once per layer" if absorbed else "cached K / V"}
    if absorbed:
        # the figure of the earlier rounds (SURVEY.md 8d bytes: cached K AND V of every layer) over the same time, so the rounds compare
        kv_total = total + cross
        out["cached_kv_accounting"] = {"bytes_per_step": int(kv_total), "achieved": round(kv_total / (ms * 1e-3) / 1e9, 1),
                                       "frac": round(kv_total / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                       "note": "the step now reads half of these bytes; frac above counts what it actually reads"}
    return out


def cpu_baseline(n_clips: int = 1):
    """The CPU oracle (a torch-CPU port of the reference semantics; the reference's own MLX path
    cannot run here) on a bounded sample of the same workload: n_clips clips, full pipeline.
    Returns (baseline dict, the oracle's GreedyResult with its per-step logits, the oracle's encoder features) -- the ids
    and logits feed `parity_vs_cpu`."""
    from oracle import whisper_ref as R

    cores = host_cores()
    torch.set_num_threads(cores)
    dims = R.DIMS["small"]
    W = R.synthetic_weights(dims, seed=0)
    audio = np.stack([R.synthetic_clip(i, 30.0) for i in range(n_clips)])
    sp = R.SpecialTokens.multilingual()
    always, first = R.suppress_lists(sp)
    t0 = time.time()
    with torch.no_grad():
        mels = np.stack([R.log_mel_spectrogram(a) for a in audio])
        xa = R.encoder_forward(W, dims, torch.from_numpy(mels))
        ref = R.greedy_decode(W, dims, xa, sp.sot_sequence_including_notimestamps(0), always, first, sp.eot,
                              sample_len=NEW_TOKENS, stop_on_eot=False, keep_logits=True)
    dt = time.time() - t0
    return {"value": round(n_clips * 30.0 / dt, 2), "unit": "audio-s/s", "cores": cores, "kind": "port",
            "sample": f"{n_clips} clip(s) x 30 s, whisper-small fp32, mel+encoder+{NEW_TOKENS} greedy steps, torch-CPU oracle, {dt:.1f} s"}, ref, xa


def cpu_oracle_peaky(xa):
    """the oracle's greedy ids for the "peaky" preset on the features it already computed (the preset only changes the
    decoder's positional table, so the encoder output is the same)"""
    from oracle import whisper_ref as R

    dims = R.DIMS["small"]
    W = R.synthetic_weights(dims, seed=0, preset="peaky")
    sp = R.SpecialTokens.multilingual()
    always, first = R.suppress_lists(sp)
    with torch.no_grad():
        return R.greedy_decode(W, dims, xa, sp.sot_sequence_including_notimestamps(0), always, first, sp.eot,
                               sample_len=NEW_TOKENS, stop_on_eot=False, keep_logits=True)


def gpu_features(model, audio_dev):
    from whisper_ipa_amd import audio as A
    from whisper_ipa_amd.runtime import on_stream

    with on_stream():
        mel = A.log_mel_padded(audio_dev, model.dims.n_mels, model.dtype)
        return model.encode_padded(mel, audio_dev.shape[0])


def parity_vs_cpu(model, audio_dev, gpu_tokens: np.ndarray, ref, setup, preset: str):
    """The bench's own token ids (bf16 on the GPU) against the CPU oracle's (f32) for the same clips and weights: overall
    match rate, the matching prefix (ids after a row's first divergence follow another history), per clip the first differing
    step with the oracle's top-1 margin there -- and the MEASURED logit error that explains it: the decode-step path is driven
    along the oracle's token history (decoding.forced_decode_logits) and err[b, s] = max_v |logit_gpu - logit_oracle|; two
    logit vectors within e of each other can only rank candidates differently that are <= 2e apart, so a first divergence
    is `explained` when oracle_margin <= 2 * err at that (row, step).  No adjustable gate."""
    from whisper_ipa_amd.decoding import forced_decode_logits

    init, always, first, eot = setup
    n_init = len(init)
    n = ref.tokens.shape[0]
    got, want = gpu_tokens[:n, n_init:], ref.tokens[:, n_init:]
    steps = min(got.shape[1], want.shape[1])
    eq = got[:, :steps] == want[:, :steps]
    trace, chosen = forced_decode_logits(model, gpu_features(model, audio_dev[:n]), ref.tokens, n_init, always, first, eot)
    err = np.zeros(ref.margins.shape)
    for b in range(n):
        ok = np.isfinite(ref.step_logits[b])
        err[b] = np.where(ok, np.abs(trace[b].cpu().numpy() - np.where(ok, ref.step_logits[b], 0.0)), 0.0).max(axis=1)
    spread = float(ref.step_logits[np.isfinite(ref.step_logits)].std())
    firsts, prefix, explained = [], 0, True
    for b in range(n):
        bad = np.flatnonzero(~eq[b])
        if bad.size:
            s0 = int(bad[0])
            ok = bool(ref.margins[b, s0] <= 2.0 * err[b, s0])
            explained = explained and ok
            firsts.append({"clip": b, "step": s0, "oracle_margin": round(float(ref.margins[b, s0]), 5),
                           "logit_err_there": round(float(err[b, s0]), 5), "explained": ok})
            prefix += s0
        else:
            prefix += steps
    flips = chosen != ref.tokens[:, n_init:]
    return {"preset": preset, "clips": int(n), "steps": int(steps), "token_match": round(float(eq.mean()), 4),
            "prefix_match": round(prefix / float(eq.size), 4), "rows_identical": int(eq.all(axis=1).sum()),
            "first_divergence": firsts, "divergences_explained_by_logit_err": explained,
            "max_logit_err": round(float(err.max()), 5), "logit_std": round(spread, 4), "rel_logit_err": round(float(err.max()) / spread, 5),
            "teacher_forced_choices_differing": int(flips.sum()),
            "teacher_forced_flips_explained": bool((ref.margins[flips] <= 2.0 * err[flips]).all()),
            "oracle_min_margin": round(float(ref.margins.min()), 5), "oracle_median_margin": round(float(np.median(ref.margins)), 4),
            "note": "GPU path bf16 vs CPU oracle f32; the f32 GPU path is bit-exact at this depth (tests/test_gpu_full_depth.py)"}


# --------------------------------------------------------------------------------------------------------- multi-rank
def visible_gpu_count():
    """GPUs this process would see, WITHOUT touching HIP (the launcher parent must stay GPU-free: its children are started
    after it, and torch.cuda.device_count() can fall through to hipGetDeviceCount on builds without amdsmi): the KFD topology
    in sysfs (GPU nodes have simd_count > 0), narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES.  None when unknown."""
    import glob

    nodes = 0
    try:
        for prop in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            for line in open(prop):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    nodes += 1
    except Exception:
        return None
    if nodes == 0:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            ids = [x for x in v.split(",") if x.strip() != ""]
            nodes = min(nodes, len(ids))
    return nodes


def launch_ranks(n: int, argv, timeout_s: float = 3000.0) -> int:
    """--gpus N without a launcher: start N rank processes of this script (one per GPU) and relay rank 0's stdout.
    This parent never initialises HIP (children are fresh processes, no exec after GPU init anywhere).  Every child is
    watched: when one exits non-zero (or the overall timeout passes) the others are terminated and that code is returned --
    a rank that died before the rendezvous must not leave the others waiting in init_process_group with the GPU held."""
    import socket
    import subprocess
    import tempfile

    n_dev = visible_gpu_count()
    share = os.environ.get("WIPA_BENCH_SHARE_GPU") == "1"
    if n_dev is not None and n_dev < n and not share:
        print(f"bench.py: --gpus {n} but only {n_dev} GPU(s) visible (WIPA_BENCH_SHARE_GPU=1 rehearses on one GPU over gloo)",
              file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    deadline = time.time() + timeout_s
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        failed = [c for c in codes if c not in (None, 0)]
        if failed or time.time() > deadline:
            rc = abs(failed[0]) if failed else 124
            print(f"bench.py: {'a rank exited with ' + str(failed[0]) if failed else 'timeout'}: terminating the other ranks", file=sys.stderr)
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    out0.seek(0)
    out = out0.read()
    # stdout carries the JSON line only: anything else a rank's libraries printed there (gloo's connection banner in the
    # one-GPU rehearsal) goes to stderr
    for line in out.decode().splitlines():
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    return rc


def init_ranks(args):
    """(rank, world, dist-or-None, device index).  Asserts that the world is what --gpus says and, on real multi-GPU
    runs, that every rank sits on its own device."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}: launch with --nproc-per-node {args.gpus}"
    # Rehearsal on a one-GPU box: WIPA_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo for the timing
    # collectives (RCCL refuses two ranks on one device); real runs use one GPU per rank and RCCL.
    share_gpu = os.environ.get("WIPA_BENCH_SHARE_GPU") == "1"
    device_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        assert dist.get_world_size() == args.gpus
        if not share_gpu:
            ids = [None] * world
            props = torch.cuda.get_device_properties(device_index)
            dist.all_gather_object(ids, (os.uname().nodename, device_index, str(getattr(props, "uuid", ""))))
            assert len(set(ids)) == world, f"ranks share a device: {ids}"
    return rank, world, dist, device_index


def timed_barrier(dist):
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(dist, elapsed: float) -> float:
    if dist is None:
        return elapsed
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# --------------------------------------------------------------------------------------------------------- fine-tune
def run_train(args):
    """--mode train: the step the reference times (scripts/train_whisper_ipa.py:552-555 -> train_step :266-311): frozen
    encoder forward, teacher-forced decoder, masked CE, backward w.r.t. the decoder, per-tensor clip, AdamW -- float32 as
    the reference trains (:505), one rank's share of BASELINE.json configs[2] (32 clips, 64 target tokens) per GPU; under
    N > 1 the decoder gradients (614 MB) are all-reduced per block over RCCL, overlapped with the backward."""
    rank, world, dist, _ = init_ranks(args)
    out = measure_train(args, rank, world, dist, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def synthetic_train_batch(B: int, T: int, n_mels: int, rank: int = 0):
    """(mel [B, 3000, n_mels] f32, token rows [B, T + 1] int64, eot): the seeded batch `--mode train` steps on (host
    tensors).  Rows are SOT-framed and EOT-padded to ragged lengths like IPADataset._tokenize_ipa_batch
    (scripts/ipa_data_loader.py:102-131); tests/test_gpu_training.py feeds the same batch to the CPU oracle."""
    g = torch.Generator().manual_seed(1000 + rank)
    mel = torch.randn(B, 3000, n_mels, generator=g) * 0.5
    eot = 50257
    tok = torch.randint(0, 50000, (B, T + 1), generator=g)
    tok[:, :4] = torch.tensor([50258, 50259, 50359, 50363])
    for b in range(B):  # ragged targets: EOT-padded tails like ipa_data_loader.py:124-131
        tok[b, T + 1 - (b % 9):] = eot
    tok[:, -1] = eot
    return mel, tok, eot


def measure_train(args, rank, world, dist, steps, warmup):
    """the timed fine-tune steps; returns the JSON object on rank 0 (None elsewhere)"""
    from whisper_ipa_amd.training import DecoderTrainer
    from whisper_ipa_amd.whisper import Whisper

    B, T = args.train_batch, args.train_tokens
    dims, W = synthetic_weights_small(0, args.model)
    model = Whisper(dims, dtype=torch.float32, f32_split=(args.f32 == "split"))
    model.load_weights(W)
    del W
    tr = DecoderTrainer(model, lr=1e-5)
    mel, tok, eot = synthetic_train_batch(B, T, dims.n_mels, rank)
    mel, tok = mel.cuda(), tok.cuda()
    log(f"train: rank {rank}/{world}, {B} clips x {T} target tokens, f32 products: {args.f32}")
    for _ in range(max(1, warmup)):
        loss, _ = tr.train_step(mel, tok, eot)
    timed_barrier(dist)
    t0 = time.perf_counter()
    exposed = 0.0
    for _ in range(steps):
        loss, _ = tr.train_step(mel, tok, eot)
        exposed += tr.last_allreduce_exposed_ms
    timed_barrier(dist)
    elapsed = max_over_ranks(dist, time.perf_counter() - t0)
    # The same step with the frozen encoder's output kept in HBM per clip (training.FrozenFeatureCache; bit-identical features,
    # tests/test_gpu_training.py): what a fine-tune run costs from the second visit of a clip on.  Reported NEXT TO the
    # headline, which stays the reference's recompute-every-step semantics (scripts/train_whisper_ipa.py:223).
    cache = tr.enable_feature_cache(B)
    keys = list(range(rank * B, rank * B + B))
    tr.train_step(mel, tok, eot, clip_keys=keys)  # first visit: computes and stores the features
    timed_barrier(dist)
    t1 = time.perf_counter()
    for _ in range(steps):
        loss_c, _ = tr.train_step(None, tok, eot, clip_keys=keys)
    timed_barrier(dist)
    elapsed_cached = max_over_ranks(dist, time.perf_counter() - t1)
    assert cache.misses == B and cache.hits == B * steps
    tr.feature_cache = None
    del cache
    # stage split (rank 0, after the timed region): encoder / loss+grads / update, median of 3
    stages = {}
    if rank == 0 or dist is not None:
        for name in ("encoder", "loss_and_grads", "update"):
            ts = []
            for _ in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                if name == "encoder":
                    feats = model.embed_audio(mel)
                elif name == "loss_and_grads":
                    tr.loss_and_grads(feats, tok, eot)
                else:
                    tr.apply_update()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t1) * 1e3)
            stages[name + "_ms"] = round(sorted(ts)[1], 2)
    if rank == 0:
        d, L, V, Ta = dims.n_text_state, dims.n_text_layer, dims.n_vocab, dims.n_audio_ctx
        M = B * T
        # dense contractions of the step: decoder fwd + dgrad + wgrad (3x) on M token rows, cross K/V projection fwd + wgrad
        # (2x, no dgrad: the encoder is frozen) on B*1500 rows, logits 3x; encoder forward once
        dec = 3 * 2.0 * M * L * (4 * d * d + 2 * d * d + 8 * d * d) + 3 * 2.0 * M * V * d
        ckv = 2 * 2.0 * B * Ta * L * 2 * d * d
        de = dims.n_audio_state
        enc = B * (2.0 * (3000 * de * 3 * dims.n_mels + Ta * de * 3 * de + dims.n_audio_layer * Ta * 12 * de * de)
                   + dims.n_audio_layer * 4.0 * Ta * Ta * de)
        # exact: the f32 MFMA (MI355X_MICROARCH.md: 157.3 TFLOP/s).  split: every f32 product is THREE bf16 MFMA terms, so the peak of
        # f32-equivalent FLOPs is a third of the dense bf16 peak -- never the f32 MFMA peak for work that runs on the bf16 MFMA
        f32_peak = 157.3 if args.f32 == "exact" else MFMA_BF16_PEAK_TFLOPS / 3.0
        ms = 1000.0 * elapsed / steps
        out = {
            "metric": f"fine-tune clips/sec (whisper-{args.model} decoder-only, f32, {B} clips x {T} tokens per GPU)",
            "value": round(world * B * steps / elapsed, 1), "unit": "clips/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.f32 == "exact" else "f32 storage and accumulation; products as three split-bf16 MFMA terms (~5e-6 relative)",
            "data": f"synthetic (seeded mel + token rows, random-init whisper-{args.model} weights)",
            "config": {"workload": f"whisper-{args.model} decoder fine-tune step (frozen encoder fwd + decoder fwd/bwd + masked CE + "
                                   f"clip + AdamW), {B} clips x 30 s, {T} target tokens per GPU",
                       "clips_per_gpu": B, "target_tokens": T, "f32_products": args.f32,
                       "clip_scope": f"{tr.clip_scope} (train_whisper_ipa.py:287-303 as written: per-tensor clip of the tensors clip_grad_dict reaches)",
                       "parallelism": f"dp{world} (per-block async all-reduce of 614 MB decoder gradients)" if world > 1 else "dp1"},
            "loss": round(float(loss), 4),
            "stages": stages,
            "roofline": {"kernel": "f32 GEMM set of the step (decoder fwd/dgrad/wgrad + cross-K/V + encoder)", "bound": "mfma",
                         "achieved": round((dec + ckv + enc) / (ms * 1e-3) / 1e12, 1), "peak": round(f32_peak, 1), "unit": "TFLOP/s",
                         "frac": round((dec + ckv + enc) / (ms * 1e-3) / 1e12 / f32_peak, 4), "traffic": None,
                         "note": "whole-step f32-equivalent FLOPs / whole-step time (includes attention backward, CE, optimiser); peak = "
                                 + ("the f32 MFMA's" if args.f32 == "exact" else "a third of the dense bf16 MFMA peak (three bf16 terms per f32 product)")},
            "with_feature_cache": {"what": "the same step with each clip's frozen-encoder output cached in HBM (bit-identical features; "
                                           "every clip seen before)", "ms_per_step": round(1000.0 * elapsed_cached / steps, 2),
                                   "value": round(world * B * steps / elapsed_cached, 1), "unit": "clips/s",
                                   "cache_bytes_per_clip": Ta * d * 4},
            "allreduce_exposed_ms_per_step": round(exposed / steps, 3) if world > 1 else 0.0,
            "grad_bytes": tr.n_params * 4,
        }
        return out
    return None


def build_model(name: str, dtype: str = "bf16", weights: str = "bf16", activations: str = "bf16", cross_attention: str = "auto",
                f32: str = "exact"):
    """random-init whisper-<name> on the GPU (seed 0; no checkpoint exists offline), optionally with fp8 e4m3 weights / activations"""
    from whisper_ipa_amd.whisper import Whisper

    if activations == "fp8" and weights != "fp8":
        sys.exit("bench.py: --activations fp8 needs --weights fp8")
    dims, W = synthetic_weights_small(0, name)
    log(f"whisper-{name} weights generated")
    model = Whisper(dims, dtype=torch.bfloat16 if dtype == "bf16" else torch.float32, f32_split=(f32 == "split"),
                    cross_attention=cross_attention)
    model.load_weights(W)
    del W
    if weights == "fp8":
        model.quantize_weights("fp8_e4m3", activations=activations)
        log(f"weights quantised to fp8 e4m3 (encoder activations: {activations})")
    return model


def run_phase(args, model, audio_dev, opts, bench_splits: int, rank: int, dist) -> None:
    """DIAGNOSTIC (--phase enc | dec): one half of the pass alone with the same passes in flight -- never a benchmark number"""
    from whisper_ipa_amd.pipeline import TranscribePipeline

    P = args.pipeline
    if args.phase == "enc":
        phase_enc_loop(model, audio_dev, P, P)
        torch.cuda.synchronize()
        timed_barrier(dist)
        t0 = time.perf_counter()
        phase_enc_loop(model, audio_dev, args.steps, P)
    else:
        feats = [gpu_features(model, audio_dev) for _ in range(P)]  # one stored encoder output per pass in flight
        torch.cuda.synchronize()
        with TranscribePipeline(model, opts, P, max_new_tokens=NEW_TOKENS, stop_on_eot=False, cross_splits=bench_splits) as pipe:
            for i in range(P):
                pipe.submit(feats[i])
            for _ in pipe.drain():
                pass
            torch.cuda.synchronize()
            timed_barrier(dist)
            t0 = time.perf_counter()
            for i in range(args.steps):
                pipe.submit(feats[i % P])
            for _ in pipe.drain():
                pass
    timed_barrier(dist)
    elapsed = max_over_ranks(dist, time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({"diagnostic_phase": args.phase, "ms_per_pass": round(1000.0 * elapsed / args.steps, 2), "passes_in_flight": P,
                          "cross_frame_splits": bench_splits or 4, "note": "one half of the pass only: NOT a benchmark number"}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def evaluate_style(model, audio_dev, args, world: int, value: float):
    """The headline workload the way the reference's batch caller consumes it (scripts/evaluate_model.py:127-232, here
    scripts/evaluate_model.py -> pipeline.transcribe_batches): DecodingOptions(language="en", without_timestamps=True,
    sample_len=NEW_TOKENS), EARLY STOP ON (the decode steps go out in chunks of 8 with an EOT probe behind each; the random-init
    model never ends a row, so the work equals the timed region's), every row turned into text on the host, results consumed
    in order.  Audio resident on the GPU, as in the timed region."""
    from whisper_ipa_amd.decoding import DecodingOptions
    from whisper_ipa_amd.pipeline import transcribe_batches

    B, P = audio_dev.shape[0], args.pipeline
    opts = DecodingOptions(language="en", without_timestamps=True, sample_len=NEW_TOKENS)
    kw = dict(passes_in_flight=P, cross_splits=args.cross_splits)
    for _ in transcribe_batches(model, [audio_dev] * P, opts, **kw):  # the chunked calls replay the graphs captured above
        pass
    torch.cuda.synchronize()
    n = max(args.steps, 2 * P)
    rows = steps = 0
    t0 = time.perf_counter()
    for r in transcribe_batches(model, (audio_dev for _ in range(n)), opts, **kw):
        rows += len(r.texts)
        steps += r.n_steps
    dt = time.perf_counter() - t0
    v = world * B * 30.0 * n / dt
    return {"what": "scripts/evaluate_model.py's loop on resident audio: transcribe_batches(model, batches, DecodingOptions(language='en', "
                    f"without_timestamps=True, sample_len={NEW_TOKENS}), passes_in_flight={P}), early stop armed (EOT probes every 8 steps), "
                    "ids -> text on the host for every row",
            "value": round(v, 1), "unit": "audio-s/s", "ms_per_batch": round(1e3 * dt / n, 2), "batches": n, "rows_transcribed": rows,
            "decode_steps_per_batch": steps // n, "frac_of_value": round(v / value, 4)}


def other_configs(args):
    """BASELINE.json configs[3] and configs[4] through the same product path, next to the headline (outside its timed region):
    whisper-medium bf16, 256 clips per batch, and whisper-large-v3 with fp8 e4m3 weights + fp8 x fp8 encoder GEMMs, 128 clips per
    batch; THREE batches in flight each (profiles/r05_other_configs_sweep.txt: medium 910 / 830 / 803 / 809 ms per batch with 1 / 2 /
    3 / 4 in flight, large-v3 824 / 754 / 744 / 742 -- and 82 / 152 GiB of HBM at three), NEW_TOKENS = 64 positions, fixed length.
    Each entry carries its own rooflines."""
    import gc

    res = {}
    for key, name, B, weights, acts in (("configs[3]", "medium", 256, "bf16", "bf16"), ("configs[4]", "large-v3", 128, "fp8", "fp8")):
        entry = optional_leg(res, key, lambda: one_other_config(key, name, B, weights, acts))
        if entry is not None:
            res[key] = entry
        gc.collect()
        torch.cuda.empty_cache()
    return res


def one_other_config(key: str, name: str, B: int, weights: str, acts: str):
    """one entry of `other_configs`: the model built, P passes in flight through the product path, its rooflines"""
    from whisper_ipa_amd.pipeline import TranscribePipeline

    t_cfg = time.perf_counter()
    model = build_model(name, "bf16", weights, acts)
    audio = torch.from_numpy(synthetic_audio(0, B)).cuda()
    P, steps = 3, 6
    with TranscribePipeline(model, bench_options(), P, max_new_tokens=NEW_TOKENS, stop_on_eot=False) as pipe:
        for _ in range(P):
            pipe.submit(audio)
        warm = [r.tokens for r in pipe.drain()]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = []
        for _ in range(steps):
            got += [r.tokens for r in pipe.submit(audio)]
        got += [r.tokens for r in pipe.drain()]
        dt = time.perf_counter() - t0
        splits = model.cross_splits
    absorbed = bench_absorbed(model, B)
    model.cross_splits = splits if absorbed else 0
    one_pass(model, audio)  # stream 0's state in the timed setting for the step roofline
    entry = {
        "metric": f"audio-seconds/sec transcribed (whisper-{name}, 30s clips)", "value": round(B * 30.0 * steps / dt, 1), "unit": "audio-s/s",
        "ms_per_step": round(1e3 * dt / steps, 2), "steps": steps, "n_gpus": 1, "dtype": "bf16",
        "config": {"workload": f"whisper-{name} bf16{' (fp8 e4m3 weights, fp8 x fp8 encoder GEMMs)' if weights == 'fp8' else ''} batched inference, "
                               f"batch={B}x30s synthetic clips, log-mel + encoder{'' if absorbed else ' + cross-K/V projection'} + {NEW_TOKENS} greedy "
                               f"KV-cached decode steps; {P} such passes in flight",
                   "weights": model.weights_format, "encoder_activations": model.activations_format,
                   "cross_attention": "absorbed" if absorbed else "cached", "clips_per_gpu": B, "passes_in_flight": P,
                   "cross_frame_splits": (splits or 4) if absorbed else None},
        "passes_identical": bool(all((t == warm[0]).all() for t in got + warm)),
        "roofline": roofline_cross_attn(model, B, iters=24),
        "decode_step": decode_step_roofline(model, B, n_steps=16),
    }
    if B <= 128:
        entry["roofline_mfma"] = roofline_mfma(model, audio)
    del model, audio, pipe
    log(f"other_configs {key}: {entry['ms_per_step']} ms per {B}-clip pass ({time.perf_counter() - t_cfg:.1f} s incl. weights)")
    return entry


def optional_leg(out: dict, name: str, fn):
    """Run one OPTIONAL leg of the line (everything but the headline, its id checks, `roofline`, `decode_step` and `cpu_baseline`, which
    stay hard failures): an exception -- e.g. out of memory in the large-v3 leg -- is logged with its traceback and recorded under
    `leg_errors` instead of costing the whole line, headline included.  Returns fn()'s value or None."""
    import traceback

    try:
        return fn()
    except Exception as e:  # noqa: BLE001 -- the point is to keep the measured headline
        log(f"OPTIONAL LEG '{name}' FAILED: {e!r}")
        traceback.print_exc(file=sys.stderr)
        out.setdefault("leg_errors", {})[name] = repr(e)[:500]
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
        return None


def main():
    global NEW_TOKENS
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", default="infer", choices=["infer", "train"],
                    help="infer: the headline metric (BASELINE.json); train: the decoder fine-tune step (configs[2] share)")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--pipeline", type=int, default=N_PIPELINE,
                    help="consecutive passes kept in flight: whisper_ipa_amd.pipeline.transcribe_batches(passes_in_flight=...)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-finetune", action="store_true", help="skip the short fine-tune step measurement appended to the default line")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the one-pass measurements of BASELINE.json configs[3] (whisper-medium, 256 clips) and configs[4] "
                         "(whisper-large-v3 fp8, 128 clips) appended to the default line")
    ap.add_argument("--new-tokens", type=int, default=NEW_TOKENS,
                    help="decode positions per clip: 64 is the benchmark setting (SURVEY.md 8d), 224 the reference's cap (secondary)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"],
                    help="sizing runs only: the benchmark metric is quoted in bf16 (f32 is what the reference's scripts set)")
    ap.add_argument("--model", default="small", choices=["tiny", "base", "small", "medium", "large-v3"],
                    help="sizing runs only: the benchmark metric is quoted on whisper-small")
    ap.add_argument("--weights", default="bf16", choices=["bf16", "fp8"],
                    help="fp8: BASELINE.json configs[4] sizing runs (e4m3 weights with per-row scales, bf16 activations); the "
                         "benchmark metric is quoted on bf16 weights")
    ap.add_argument("--cross-attention", default="auto", choices=["auto", "cached", "absorbed"],
                    help="decode-step cross-attention: the encoder output with absorbed key / value projections (what auto picks for "
                         "bf16 models of <= 16 heads) or mlx_whisper's projected K / V caches (Whisper(cross_attention=...))")
    ap.add_argument("--cross-splits", type=int, default=None, choices=[0, 1, 2, 3, 4],
                    help="absorbed cross-attention: frame splits per clip of the decode step's streaming launch.  Default: what "
                         "pipeline.transcribe_batches sets -- 2 (half-chip launches) with >= 2 passes in flight, else the "
                         "library default 4 (shortest lone step)")
    ap.add_argument("--activations", default="bf16", choices=["bf16", "fp8"],
                    help="with --weights fp8: fp8 also runs the encoder's q|k, value, mlp1, mlp2 projections fp8 x fp8 on the "
                         "block-scaled fp8 MFMA (LayerNorm / GELU outputs quantised per row) -- configs[4] '(CDNA4 fp8 MFMA)'")
    ap.add_argument("--f32", default="exact", choices=["exact", "split"],
                    help="float32 runs: exact f32 MFMA products (the bench's choice, as the reference computes: a benchmarked float32 figure is "
                         "never an emulated one) or three split-bf16 MFMA terms per product (what Whisper() sets for float32 since round 5)")
    ap.add_argument("--phase", default="", choices=["", "enc", "dec"],
                    help="DIAGNOSTIC: time only log-mel + encoder (enc) or only the decode loops on stored features (dec) with "
                         "the same passes in flight; prints ms per pass and exits -- not a benchmark line")
    ap.add_argument("--train-batch", type=int, default=32)
    ap.add_argument("--train-tokens", type=int, default=64)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    NEW_TOKENS = args.new_tokens
    torch.set_num_threads(rank_threads())  # cores / LOCAL_WORLD_SIZE: eight ranks must not ask for 8 x 16 host threads
    if args.mode == "train":
        return run_train(args)
    rank, world, dist, device_index = init_ranks(args)

    from whisper_ipa_amd.pipeline import PIPELINE_CROSS_SPLITS, TranscribePipeline, transcribe_batches
    from whisper_ipa_amd.runtime import hw_queues

    B = args.batch
    log(f"start: rank {rank}/{world}, host cores {host_cores()}, torch threads of this rank {torch.get_num_threads()}, "
        f"hardware queues {hw_queues()}")
    model = build_model(args.model, args.dtype, args.weights, args.activations, args.cross_attention, args.f32)
    # the streaming-launch setting of the timed region: the pipeline's own rule unless --cross-splits overrides it
    bench_splits = args.cross_splits if args.cross_splits is not None else (PIPELINE_CROSS_SPLITS if args.pipeline >= 2 else 0)
    audio_dev = torch.from_numpy(synthetic_audio(rank * B, B)).cuda()
    setup = decode_setup(model.num_languages)
    opts = bench_options()
    bench_packed(model, B)
    torch.cuda.synchronize()
    log("model + audio resident on the GPU")

    if args.phase:
        return run_phase(args, model, audio_dev, opts, bench_splits, rank, dist)

    # THE TIMED REGION IS THE PRODUCT PATH: whisper_ipa_amd.pipeline.TranscribePipeline -- the scheduler behind
    # transcribe_batches(model, batches, options, passes_in_flight=P), which scripts/evaluate_model.py and validate() call.
    # Pass i runs on stream set (i % P) and is only collected when its stream set is needed again, so the encoder of pass
    # i + 1 overlaps the decode loop of pass i; each pass does all of its work inside the timed region on its own workspaces /
    # KV caches.  stop_on_eot=False: a fixed NEW_TOKENS positions per clip (EOT latch on), so the work is fixed.
    collected = []  # every timed pass's ids (host arrays), compared after the timed region
    with TranscribePipeline(model, opts, args.pipeline, max_new_tokens=NEW_TOKENS, stop_on_eot=False, cross_splits=bench_splits) as pipe:
        for i in range(args.warmup):
            for _ in range(args.pipeline):  # warm every stream set (workspaces, KV caches, captured graphs)
                pipe.submit(audio_dev)
            for _ in pipe.drain():
                pass
            torch.cuda.synchronize()
            log(f"warmup pass {i} done")
        timed_barrier(dist)
        t0 = time.perf_counter()
        for i in range(args.steps):
            collected += [r.tokens for r in pipe.submit(audio_dev)]
        collected += [r.tokens for r in pipe.drain()]
        timed_barrier(dist)
        elapsed = max_over_ranks(dist, time.perf_counter() - t0)
    tokens = collected[-1]
    assert len(collected) == args.steps
    model.cross_splits = bench_splits  # the legs below measure the setting that was timed

    out = None
    absorbed_run = bench_absorbed(model, B)
    if rank == 0:
        audio_seconds = world * B * 30.0 * args.steps
        out = {
            "metric": f"audio-seconds/sec transcribed (whisper-{args.model}, 30s clips)",
            "value": round(audio_seconds / elapsed, 1),
            "unit": "audio-s/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1000.0 * elapsed / args.steps, 2),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": f"synthetic (seeded noise clips, random-init whisper-{args.model} weights)",
            "config": {"weights": model.weights_format, "encoder_activations": model.activations_format,
                       "cross_attention": "absorbed" if absorbed_run else "cached",
                       # what a pass really runs: with the absorbed form there is NO cross-K/V projection (the decode steps stream
                       # the encoder output itself); with cached K / V the projection of every decoder layer is part of the pass
                       "workload": f"whisper-{args.model} {args.dtype}{' (fp8 e4m3 weights)' if args.weights == 'fp8' else ''} batched inference, batch={B}x30s synthetic clips per GPU, "
                                   f"log-mel + encoder{'' if absorbed_run else ' + cross-K/V projection'} + {NEW_TOKENS} greedy KV-cached decode steps"
                                   f" ({'cross-attention on the encoder output, key / value projections absorbed' if absorbed_run else 'cross-attention on cached K / V'});"
                                   f" {args.pipeline} such passes ({args.pipeline * B} clips) in flight per GPU"
                                   + (f"; streaming launch in {model.cross_splits} frame splits per clip (Whisper.cross_splits: the setting for several "
                                      f"passes in flight; 4 = the lone-decode default, timed below as *_default_splits)"
                                      if absorbed_run and model.cross_splits not in (0, 4) else ""),
                       "cross_frame_splits": (model.cross_splits or 4) if absorbed_run else None,
                       "clips_per_gpu": B, "clips_in_flight_per_gpu": args.pipeline * B, "new_tokens": NEW_TOKENS,
                       "passes_in_flight": args.pipeline,
                       "schedule": "whisper_ipa_amd.pipeline.TranscribePipeline (the scheduler of transcribe_batches; what "
                                   "scripts/evaluate_model.py and validate() call)",
                       "hw_queues": hw_queues(),
                       "parallelism": f"dp{world} (clip sharding, no collective)"},
            "tokens_checksum": int(tokens.sum() % 1000003),
            # every timed pass transcribes the same clips: identical ids in all of them (a race in a kernel would show here)
            "passes_identical": bool(all(t.shape == collected[0].shape and (t == collected[0]).all() for t in collected)),
        }
        log(f"timed region done: {elapsed:.3f} s for {args.steps} passes")
        # ---- everything below is OUTSIDE the timed region, one pass in flight
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        single = one_pass(model, audio_dev)
        out["ms_per_pass_single_in_flight"] = round((time.perf_counter() - t1) * 1e3, 2)
        # the same metric with ONE pass (B clips) resident at a time: the latency figure next to the throughput figure `value`
        out["value_single_in_flight"] = round(world * B * 30.0 / (out["ms_per_pass_single_in_flight"] * 1e-3), 1)
        assert (single == tokens).all(), "the pipelined and the single pass disagree on the ids"
        assert out["passes_identical"], "timed passes over the same clips produced different ids"
        ev = optional_leg(out, "evaluate_style", lambda: evaluate_style(model, audio_dev, args, world, out["value"]))
        if ev is not None:
            out["evaluate_style"] = ev
        log("evaluate-style run done")
        one_pass(model, audio_dev)  # stream 0's decode state back to the timed setting (evaluate_style re-captured nothing else)
        out["roofline"] = roofline_cross_attn(model, B)
        out["decode_step"] = decode_step_roofline(model, B)
        if absorbed_run and model.cross_splits not in (0, 4):
            # the same two latency figures in the library's DEFAULT setting (4 frame splits: what a caller with one pass at a
            # time runs); the bench setting trades them for throughput with several passes in flight
            keep = model.cross_splits
            model.cross_splits = 0
            one_pass(model, audio_dev)  # captures the step graph of this setting
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            one_pass(model, audio_dev)
            ms1 = (time.perf_counter() - t2) * 1e3
            dflt = decode_step_roofline(model, B)
            rf4 = roofline_cross_attn(model, B)
            out["default_splits"] = {"cross_frame_splits": 4, "ms_per_pass_single_in_flight": round(ms1, 2),
                                     "value_single_in_flight": round(world * B * 30.0 / (ms1 * 1e-3), 1),
                                     "decode_step_ms": dflt["ms_per_step"], "decode_step_frac": dflt["frac"],
                                     "streaming_launch": {k: rf4[k] for k in ("workgroups", "avg_launch_ms", "achieved", "frac", "traffic",
                                                                              "algorithmic_bytes_per_launch")}}
            # next to the timed setting's own figures: the same kernel as the library launches it by default (all 256 CUs)
            out["roofline"]["library_default_setting"] = dict(out["default_splits"]["streaming_launch"], frame_splits=4)
            model.cross_splits = keep
            one_pass(model, audio_dev)  # back to the timed setting: the parity legs below check what was timed
        if args.dtype == "bf16" and args.batch <= 128:
            rm = optional_leg(out, "roofline_mfma", lambda: roofline_mfma(model, audio_dev))
            if rm is not None:
                out["roofline_mfma"] = rm
        log("roofline microbenches done")
        if world == 1 and not args.no_cpu_baseline and args.model == "small" and B >= 8:
            out["cpu_baseline"], ref, xa_ref = cpu_baseline(8)  # ~25 s of host work
            out["parity_vs_cpu"] = parity_vs_cpu(model, audio_dev, single, ref, setup, "lively")
            log("cpu baseline + parity done")
            if args.weights == "bf16":
                # the "peaky" preset (a confident model: top-1 margins of several logit standard deviations): here the bf16 ids
                # must be the f32 oracle's, bit for bit, for every clip and step.  Outside the timed region; the headline
                # number stays on the lively preset.
                ref_p = cpu_oracle_peaky(xa_ref)
                lively_pos = model.flat_parameters()["decoder.positional_embedding"].clone()
                _, Wp = synthetic_weights_small(0, args.model, preset="peaky")
                model.load_weights({"decoder.positional_embedding": Wp["decoder.positional_embedding"]}, strict=False)
                del Wp
                peaky_ids = one_pass(model, audio_dev[:8].contiguous())
                out["parity_vs_cpu_peaky"] = parity_vs_cpu(model, audio_dev, peaky_ids, ref_p, setup, "peaky")
                model.load_weights({"decoder.positional_embedding": lively_pos}, strict=False)
                log("peaky preset parity done")
        if world == 1 and not args.no_finetune and args.model == "small" and args.dtype == "bf16" and args.weights == "bf16":
            # BASELINE.json configs[2], one rank's share, next to the headline number (outside its timed region): the step the
            # reference times at scripts/train_whisper_ipa.py:552-555, float32 with exact products.  `--mode train` is the
            # full-length / multi-rank form of the same measurement.
            import gc

            del model
            gc.collect()  # the decode states reference the model (cycle): collect before returning the 20+ GB of caches
            torch.cuda.empty_cache()
            ft = optional_leg(out, "finetune_step", lambda: measure_train(args, 0, 1, None, steps=3, warmup=1))
            if ft is not None:
                out["finetune_step"] = {k: ft[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype", "config", "stages", "roofline", "loss",
                                                           "with_feature_cache")}
            log("fine-tune step measured")
            model = None
        if (world == 1 and not args.no_other_configs and args.model == "small" and args.dtype == "bf16" and args.weights == "bf16"
                and B == BATCH and NEW_TOKENS == 64):
            import gc

            model = None
            gc.collect()
            torch.cuda.empty_cache()
            oc = optional_leg(out, "other_configs", lambda: other_configs(args))
            if oc is not None:
                out["other_configs"] = oc
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
