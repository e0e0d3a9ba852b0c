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
Consecutive passes are software-pipelined: up to `--pipeline` (default 4) passes are in flight on
separate HIP streams with separate workspaces / KV caches, each doing ALL of its work inside the
timed region, so the encoder of one batch overlaps the decode loop of the previous one
(`ms_per_step` = timed wall time / steps, i.e. the steady-state time per 64-clip batch).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed on the launch
stream) and `cpu_baseline` (the CPU oracle = a port, timed on this box's host cores on a
bounded sample, rank 0 at N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# The pipelined passes live on several HIP streams; ROCm multiplexes streams onto GPU_MAX_HW_QUEUES hardware
# queues (default 4).  Measured r01 with 4 passes in flight: 2 queues 119.9 ms, 4 -> 108.4, 8 -> 95.4 per pass.
# Must be set before the HIP runtime starts (i.e. before torch touches the GPU).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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


BATCH = 64          # clips per GPU
N_PIPELINE = 4      # consecutive passes kept in flight on separate HIP streams (see --pipeline).  Measured r01 with
                    # GPU_MAX_HW_QUEUES=8 (ms per 64-clip pass, repeatable to 0.5 %): 1 -> 133.2, 3 -> 98.8, 4 -> 94.7,
                    # 5 -> 123.5: the MFMA-bound encoder of later passes runs in the shadows of the launch/HBM-bound
                    # decode loops of earlier ones
N_STREAMS = 1       # sub-batches of the 64 clips, one HIP stream each (measured r01: 1 -> 162 ms,
                    # 2 -> 156 ms, 4 -> 200 ms, 8 -> 266 ms per pass: the per-step cost of the decode
                    # loop is launch/latency bound and does not shrink with the sub-batch)
NEW_TOKENS = 64     # decode positions per clip (SURVEY.md section 8d primary setting)
HBM_PEAK_GBS = 8000.0
MFMA_BF16_PEAK_TFLOPS = 2500.0


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


def synthetic_weights_small(seed: int = 0, name: str = "small"):
    """Random-init whisper-<name> in mlx_whisper naming (no checkpoint exists offline).  Same
    recipe as the oracle's generator, restated here so the product path does not import it."""
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
    return dims, W


def synthetic_audio(first_clip: int, n: int) -> np.ndarray:
    """default_rng(1234 + clip), 0.1 * N(0,1) f32, 480 000 samples."""
    out = np.empty((n, 480000), dtype=np.float32)
    for i in range(n):
        out[i] = 0.1 * np.random.default_rng(1234 + first_clip + i).standard_normal(480000, dtype=np.float32)
    return out


# multilingual special ids / suppress list (tokenizer constants, whisper_ipa_amd/tokenizer.py)
def decode_setup():
    from whisper_ipa_amd.tokenizer import get_tokenizer
    from whisper_ipa_amd.decoding import DecodingOptions, _suppress_lists

    tok = get_tokenizer(True)
    always, first = _suppress_lists(DecodingOptions(language="en", without_timestamps=True), tok)
    return list(tok.sot_sequence_including_notimestamps), always, first, tok.eot


def one_pass(model, audio_chunks, setup):
    """One pass over the batch.  The clips are split into sub-batches, each on its own HIP stream:
    log-mel -> encoder -> cross-KV -> decode loop are enqueued asynchronously per sub-batch, so the
    MFMA-bound encoder of one sub-batch overlaps the HBM/latency-bound decode loop of another.
    Returns the token matrix of the whole batch (host), i.e. the pass ends when all ids are on the host."""
    return pass_collect(pass_launch(model, audio_chunks, setup, 0))


def pass_launch(model, audio_chunks, setup, stream_base: int):
    """enqueue one whole pass (asynchronously) on library streams stream_base, stream_base+1, ..."""
    from whisper_ipa_amd import audio as A
    from whisper_ipa_amd.decoding import greedy_launch
    from whisper_ipa_amd.runtime import use_stream

    init, always, first, eot = setup
    handles = []
    for sid, a in enumerate(audio_chunks):
        with use_stream(stream_base + sid):
            mel = A.log_mel_padded(a, model.dims.n_mels, model.dtype)
            feats = model.encode_padded(mel, a.shape[0])
            handles.append(greedy_launch(model, feats, init, always, first, eot, max_new_tokens=NEW_TOKENS))
    return handles


def group_launch(model, audio, setup, stream_id: int, n_batches: int):
    """enqueue n_batches passes as ONE decode group: log-mel + encoder per 64-clip batch, then one greedy loop over all
    n_batches * B clips (the decode-step projections stream the decoder weights once for the whole group)."""
    from whisper_ipa_amd import audio as A
    from whisper_ipa_amd.decoding import greedy_launch
    from whisper_ipa_amd.runtime import use_stream

    init, always, first, eot = setup
    with use_stream(stream_id):
        feats = []
        for _ in range(n_batches):
            mel = A.log_mel_padded(audio, model.dims.n_mels, model.dtype)
            feats.append(model.encode_padded(mel, audio.shape[0]))
        allf = feats[0] if n_batches == 1 else torch.cat(feats, dim=0)
        return [greedy_launch(model, allf, init, always, first, eot, max_new_tokens=NEW_TOKENS)]


def pass_collect(handles):
    from whisper_ipa_amd.decoding import greedy_collect

    return np.concatenate([greedy_collect(h).tokens for h in handles], axis=0)


def roofline_cross_attn(model, B: int, iters: int = 48):
    """Dominant HBM-bound kernel: decode-step cross-attention (K11).  Algorithmic bytes per launch
    = B * 2 * H * 1500 * 64 * sizeof(bf16) (every cached K and V element once) + q/out.
    Timed with events on the library stream over back-to-back launches that cycle through all
    layer caches (each launch streams bytes no other recent launch touched -> HBM, not cache)."""
    from whisper_ipa_amd import ops
    from whisper_ipa_amd.runtime import on_stream, stream

    d = model.dims
    H, Ta = d.n_text_head, d.n_audio_ctx
    st = model._dec_states[0]
    lay = st.layout
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
    # HBM traffic per launch from the PMC counters (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE,
    # separate rocprofv3 --pmc passes over tools/pmc_cross_attn.py; summary committed under profiles/).  Only
    # quoted when it was collected for exactly this launch shape.
    traffic = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_cross_attn.json")))
        if pm.get("algorithmic_bytes_per_launch") == bytes_alg:
            traffic = pm["hbm_bytes_per_launch"]
    except Exception:
        pass
    return {"kernel": "decode_attn_kernel (decode-step cross-attention)", "bound": "hbm", "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "algorithmic_bytes_per_launch": bytes_alg, "avg_launch_ms": round(ms, 5)}


def cpu_baseline(n_clips: int = 1):
    """The CPU oracle (a torch-CPU port of the reference semantics; the reference's own MLX path
    cannot run here) on a bounded sample of the same workload: n_clips clips, full pipeline."""
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
        R.greedy_decode(W, dims, xa, sp.sot_sequence_including_notimestamps(0), always, first, sp.eot,
                        sample_len=NEW_TOKENS, stop_on_eot=False)
    dt = time.time() - t0
    return {"value": round(n_clips * 30.0 / dt, 2), "unit": "audio-s/s", "cores": cores, "kind": "port",
            "sample": f"{n_clips} clip(s) x 30 s, whisper-small fp32, mel+encoder+{NEW_TOKENS} greedy steps, torch-CPU oracle, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--streams", type=int, default=N_STREAMS, help="clip sub-batches run on this many HIP streams")
    ap.add_argument("--pipeline", type=int, default=N_PIPELINE, help="consecutive passes kept in flight on separate HIP streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decode-group", type=int, default=1,
                    help="EXPERIMENT: decode this many consecutive 64-clip batches together (encoder still per batch)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"],
                    help="sizing runs only: the benchmark metric is quoted in bf16 (f32 is what the reference's scripts set)")
    ap.add_argument("--model", default="small", choices=["tiny", "base", "small", "medium", "large-v3"],
                    help="sizing runs only: the benchmark metric is quoted on whisper-small")
    args = ap.parse_args()

    torch.set_num_threads(host_cores())
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # Rehearsal on a one-GPU box: WIPA_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo for the two timing
    # collectives (RCCL refuses two ranks on one device); the driver's real runs use one GPU per rank and RCCL.
    share_gpu = os.environ.get("WIPA_BENCH_SHARE_GPU") == "1"
    device_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist  # RCCL: only for the timing barrier / max-reduce

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))

    from whisper_ipa_amd.whisper import Whisper

    B = args.batch
    log(f"start: rank {rank}/{world}, host cores {host_cores()}")
    dims, W = synthetic_weights_small(0, args.model)
    log("weights generated")
    model = Whisper(dims, dtype=torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    model.load_weights(W)
    del W
    audio_dev = torch.from_numpy(synthetic_audio(rank * B, B)).cuda()
    audio_chunks = [c.contiguous() for c in audio_dev.chunk(args.streams)]
    setup = decode_setup()
    model.packed()
    torch.cuda.synchronize()
    log("model + audio resident on the GPU")

    for i in range(args.warmup):
        for pset in range(args.pipeline):  # warm every stream set (workspaces, KV caches, captured graphs)
            if args.decode_group > 1:
                pass_collect(group_launch(model, audio_chunks[0], setup, pset, args.decode_group))
            else:
                pass_collect(pass_launch(model, audio_chunks, setup, pset * args.streams))
        torch.cuda.synchronize()
        log(f"warmup pass {i} done")
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # --pipeline P > 1: pass i runs on stream set (i % P) and is only collected when its stream set is needed
    # again, so the encoder of pass i+1 can overlap the decode loop of pass i (each pass still does all its work
    # inside the timed region; every pass owns separate workspaces / KV caches)
    inflight = []
    if args.decode_group > 1:
        i, gi = 0, 0
        while i < args.steps:
            g = min(args.decode_group, args.steps - i)
            if len(inflight) == args.pipeline:
                tokens = pass_collect(inflight.pop(0))
            inflight.append(group_launch(model, audio_chunks[0], setup, gi % args.pipeline, g))
            i += g
            gi += 1
    else:
        for i in range(args.steps):
            if len(inflight) == args.pipeline:
                tokens = pass_collect(inflight.pop(0))
            inflight.append(pass_launch(model, audio_chunks, setup, (i % args.pipeline) * args.streams))
    while inflight:
        tokens = pass_collect(inflight.pop(0))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
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
            "config": {"workload": f"whisper-{args.model} {args.dtype} batched inference, batch={B}x30s synthetic clips per GPU, "
                                   f"log-mel + encoder + cross-KV + {NEW_TOKENS} greedy KV-cached decode steps",
                       "clips_per_gpu": B, "new_tokens": NEW_TOKENS, "streams_per_gpu": args.streams, "passes_in_flight": args.pipeline, "decode_group": args.decode_group,
                       "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       "parallelism": f"dp{world} (clip sharding, no collective)"},
            "tokens_checksum": int(tokens.sum() % 1000003),
        }
        log(f"timed region done: {elapsed:.3f} s for {args.steps} passes")
        out["roofline"] = roofline_cross_attn(model, audio_chunks[0].shape[0])
        log("roofline microbench done")
        if world == 1 and not args.no_cpu_baseline and args.model == "small":
            out["cpu_baseline"] = cpu_baseline(8)  # ~20 s of host work
            log("cpu baseline done")
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
