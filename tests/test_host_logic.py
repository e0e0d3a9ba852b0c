"""CPU tests of the host side: tokenizer (BPE, framing, known special ids), WAV ingest,
pad_or_trim, checkpoint key conversion, batch framing, and the multi-process (gloo, world 2)
clip sharding / loss-statistics / gradient all-reduce used for N > 1."""
import base64
import json
import os
import struct
import sys
import wave

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tokenizer_special_ids_and_framing():
    from whisper_ipa_amd.tokenizer import NON_SPEECH_TOKENS_MULTI, get_tokenizer

    tok = get_tokenizer(multilingual=True)
    # WHISPER_IPA_RESEARCH_STANDALONE.md:333-338
    assert (tok.eot, tok.sot, tok.transcribe, tok.no_timestamps) == (50257, 50258, 50359, 50363)
    assert tok.to_language_token("en") == 50259 and tok.translate == 50358 and tok.timestamp_begin == 50364
    assert tok.sot_sequence_including_notimestamps == (50258, 50259, 50359, 50363)
    tok.language = "fi"  # ipa_data_loader.py:152: assignment does not change the frozen sequence
    assert tok.sot_sequence_including_notimestamps == (50258, 50259, 50359, 50363)
    assert tok.n_vocab == 51865 and len(tok.all_language_tokens) == 99
    assert tok.non_speech_tokens == NON_SPEECH_TOKENS_MULTI
    text = "n̩æp ɛ̃ tʰ əː"
    ids = tok.encode(text)
    assert tok.decode(ids) == text  # byte fallback round trip
    assert tok.decode([50258, 50259] + ids + [50257]) == "<|startoftranscript|><|en|>" + text + "<|endoftext|>"


def test_bpe_merges_with_a_rank_table(tmp_path):
    """byte-pair merging by lowest rank on a tiny synthetic .tiktoken table."""
    from whisper_ipa_amd.tokenizer import Tokenizer, load_tiktoken_ranks

    toks = [bytes([b]) for b in range(256)] + [b"ab", b"abc", b" a", "ə".encode(), "əː".encode()]
    path = tmp_path / "t.tiktoken"
    path.write_bytes(b"".join(base64.b64encode(t) + b" " + str(i).encode() + b"\n" for i, t in enumerate(toks)))
    ranks = load_tiktoken_ranks(str(path))
    tok = Tokenizer(ranks, byte_fallback=False)
    assert tok.encode("abc") == [257]
    assert tok.encode("abab") == [256, 256]
    assert tok.encode("əː") == [260]
    assert tok.eot == len(toks) and tok.sot == len(toks) + 1
    assert tok.decode(tok.encode("abc əː")) == "abc əː"


def test_suppress_lists_match_reference_options():
    from whisper_ipa_amd.decoding import DecodingOptions, _suppress_lists
    from whisper_ipa_amd.tokenizer import get_tokenizer

    tok = get_tokenizer(True)
    always, first = _suppress_lists(DecodingOptions(language="en", without_timestamps=True), tok)
    assert first == [tok.encode(" ")[0], tok.eot]
    for t in (tok.sot, tok.translate, tok.transcribe, tok.sot_lm, tok.sot_prev, tok.no_speech):
        assert t in always
    assert tok.eot not in always and tok.no_timestamps not in always
    o = DecodingOptions()
    assert o.fp16 is True and o.suppress_blank and o.suppress_tokens == "-1" and o.temperature == 0.0


def _write_wav(path, data, rate, width=2, channels=1):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(channels)
        w.setsampwidth(width)
        w.setframerate(rate)
        w.writeframes(data)


def test_load_audio_wav_and_resample(tmp_path):
    from whisper_ipa_amd.audio import load_audio, pad_or_trim

    t = np.arange(16000) / 16000.0
    x = (0.5 * np.sin(2 * np.pi * 440 * t) * 32767).astype("<i2")
    _write_wav(tmp_path / "a.wav", x.tobytes(), 16000)
    a = load_audio(str(tmp_path / "a.wav"))
    assert a.dtype == np.float32 and a.shape == (16000,)
    assert np.allclose(a, x.astype(np.float32) / 32768.0)
    st = np.stack([x, x], axis=1).reshape(-1)
    _write_wav(tmp_path / "s.wav", st.tobytes(), 16000, channels=2)
    assert np.allclose(load_audio(str(tmp_path / "s.wav")), a, atol=1e-4)
    t8 = np.arange(8000) / 8000.0
    x8 = (0.5 * np.sin(2 * np.pi * 440 * t8) * 32767).astype("<i2")
    _write_wav(tmp_path / "b.wav", x8.tobytes(), 8000)
    b = load_audio(str(tmp_path / "b.wav"))
    assert abs(len(b) - 16000) <= 1
    mid = slice(2000, 14000)
    assert np.abs(b[mid] - a[: len(b)][mid]).max() < 2e-2  # same 440 Hz tone after 8k -> 16k
    assert pad_or_trim(a).shape == (480000,) and pad_or_trim(np.zeros(500000, np.float32)).shape == (480000,)
    assert pad_or_trim(torch.zeros(2, 10), 16).shape == (2, 16)
    # 44.1 kHz -> 16 kHz (up 160 / down 441) in bounded time, tone kept, a 10 kHz component (above the new Nyquist) removed
    t44 = np.arange(3 * 44100) / 44100.0
    x44 = 0.5 * np.sin(2 * np.pi * 440 * t44) + 0.25 * np.sin(2 * np.pi * 10000 * t44)
    _write_wav(tmp_path / "c.wav", (x44 * 32767).astype("<i2").tobytes(), 44100)
    import time
    t0 = time.perf_counter()
    c = load_audio(str(tmp_path / "c.wav"))
    assert time.perf_counter() - t0 < 10.0 and abs(len(c) - 48000) <= 1
    ref = 0.5 * np.sin(2 * np.pi * 440 * np.arange(len(c)) / 16000.0)
    assert np.abs(c[2000:-2000] - ref[2000:-2000]).max() < 5e-3


def test_hf_key_conversion_roundtrip():
    from whisper_ipa_amd.load_models import convert_weights, dims_from_name
    from whisper_ipa_amd.whisper import ModelDimensions, parameter_names

    dims = ModelDimensions(80, 1500, 64, 1, 1, 51865, 448, 64, 1, 1)
    names = set(parameter_names(dims))
    hf = {
        "model.encoder.conv1.weight": torch.zeros(64, 80, 3),
        "model.encoder.layers.0.self_attn.q_proj.weight": torch.zeros(64, 64),
        "model.encoder.layers.0.self_attn.k_proj.weight": torch.zeros(64, 64),
        "model.encoder.layers.0.self_attn_layer_norm.bias": torch.zeros(64),
        "model.encoder.layers.0.fc1.weight": torch.zeros(256, 64),
        "model.encoder.layer_norm.weight": torch.zeros(64),
        "model.encoder.embed_positions.weight": torch.zeros(1500, 64),
        "model.decoder.embed_tokens.weight": torch.zeros(10, 64),
        "model.decoder.embed_positions.weight": torch.zeros(448, 64),
        "model.decoder.layers.0.encoder_attn.v_proj.bias": torch.zeros(64),
        "model.decoder.layers.0.encoder_attn_layer_norm.weight": torch.zeros(64),
        "model.decoder.layers.0.final_layer_norm.weight": torch.zeros(64),
        "model.decoder.layer_norm.bias": torch.zeros(64),
        "proj_out.weight": torch.zeros(10, 64),
    }
    out = convert_weights(hf)
    assert set(out) <= names, set(out) - names
    assert tuple(out["encoder.conv1.weight"].shape) == (64, 3, 80)
    assert "decoder.blocks.0.cross_attn.value.bias" in out and "decoder.positional_embedding" in out
    assert dims_from_name("mlx-community/whisper-small-mlx").n_audio_state == 768
    assert dims_from_name("mlx-community/whisper-large-v3-mlx").n_mels == 128
    assert len(parameter_names(ModelDimensions(80, 1500, 768, 12, 12, 51865, 448, 768, 12, 12))) == 4 + 12 * 15 + 2 + 2 + 12 * 24 + 2


def test_decoder_parameter_count_matches_reference():
    """153 580 800 trainable decoder parameters for whisper-small (benchmark_models_simple.py:52)."""
    d, L, V, ctx = 768, 12, 51865, 448
    per_block = 2 * (4 * d * d + 3 * d) + 2 * 2 * d + (d * 4 * d + 4 * d) + (4 * d * d + d) + 2 * d
    assert V * d + ctx * d + L * per_block + 2 * d == 153_580_800


def test_batch_token_framing(tmp_path):
    """ipa_data_loader.py:102-131: SOT seq + ipa + EOT, EOT padding."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import ipa_data_loader as dl
    from whisper_ipa_amd.tokenizer import get_tokenizer

    js = tmp_path / "d.json"
    js.write_text(json.dumps([{"audio_path": "x.wav", "ipa_transcription": "abc"}]))
    ds = dl.IPADataset(str(js), get_tokenizer(True))
    t = ds.tokenize_batch(["ab", "abcde"])
    assert t.dtype == torch.int32 and tuple(t.shape) == (2, 4 + 5 + 1)
    # 'a' -> 64, 'b' -> 65: Whisper's byte-level ids (reference known answer 'a' -> [64])
    assert t[0].tolist() == [50258, 50259, 50359, 50363, 64, 65, 50257, 50257, 50257, 50257]
    assert t[1].tolist()[-1] == 50257 and len(ds) == 1


def test_reference_v2_test_split_tokenizes(tmp_path):
    """the reference's text fixture data/v2_filtered/combined_test_ipa.json, if mounted, goes through
    the loader's framing verbatim (NFC-stable strings, round trip)."""
    p = "/root/reference/data/v2_filtered/combined_test_ipa.json"
    if not os.path.exists(p):
        pytest.skip("reference data not mounted")
    from whisper_ipa_amd.tokenizer import get_tokenizer
    import unicodedata

    tok = get_tokenizer(True)
    data = json.load(open(p))
    assert len(data) == 700
    for e in data[:200]:
        s = e["ipa_transcription"]
        assert unicodedata.normalize("NFC", s) == s
        assert tok.decode(tok.encode(s)) == s


# ---------------------------------------------------------------- world_size 2 on gloo
def _worker(rank, world, port, q):
    import torch.distributed as dist
    from whisper_ipa_amd import parallel as P

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows_all = [[i, i + 1, i + 2][: 1 + i % 3] for i in range(7)]
        lo, hi = P.shard_bounds(len(rows_all), world, rank)
        gathered = P.gather_token_rows(rows_all[lo:hi])
        draw = list(np.random.default_rng(0).choice(100, 8, replace=False))
        mine = P.shard_indices(draw, world, rank)
        s, n = P.allreduce_loss_stats(torch.tensor(1.5 * (rank + 1)), torch.tensor(10.0 * (rank + 1)))
        # the trainer's gradient exchange (training.py loss_and_grads): segments of ONE flat buffer reduced asynchronously in
        # the order the backward finishes them -- tail, blocks from the back, head -- then joined
        torch.manual_seed(0)
        full = torch.randn(5000)
        flat = full * (0.25 if rank == 0 else 0.75)
        red = P.SegmentReducer(flat)
        for lo, hi in [(4900, 5000), (3000, 4900), (1000, 3000), (1000, 1000), (0, 1000)]:
            red.reduce(lo, hi)
        n_joined = red.wait()
        ok = bool(torch.allclose(flat, full, atol=1e-6)) and n_joined == 4 and not red.pending
        # collective step agreement: widths are maximised; one failing rank makes EVERY rank see -1 (no lone break)
        w_ok = P.agree_on_step(17 + rank) == 18
        w_fail = P.agree_on_step(17, failed=(rank == 1)) == -1
        q.put((rank, gathered == rows_all, [int(i) for i in mine], float(s), float(n), ok and w_ok and w_fail))
    finally:
        dist.destroy_process_group()


def test_dp_sharding_and_collectives_gloo_world2():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    draw = [int(i) for i in np.random.default_rng(0).choice(100, 8, replace=False)]
    assert res[0][2] + res[1][2] == draw
    for r in res:
        assert r[1] is True and abs(r[3] - 4.5) < 1e-6 and abs(r[4] - 30.0) < 1e-6 and r[5] is True


# ---------------------------------------------------------------- world_size 8 on gloo: BASELINE configs[2]'s partition
TINY_DP = dict(n_mels=80, n_audio_ctx=12, n_audio_state=64, n_audio_head=1, n_audio_layer=1, n_vocab=96, n_text_ctx=16,
               n_text_state=64, n_text_head=1, n_text_layer=2)
TINY_EOT = 90


def _tiny_dp_problem():
    """global batch 256 of a tiny decoder (the oracle's arithmetic on 64-wide, 2-layer dims): weights, features, ragged
    EOT-padded token rows and ONE shared draw -- every rank rebuilds the same objects from the same seeds"""
    from oracle import whisper_ref as R

    dims = R.ModelDimensions(**TINY_DP)
    W = R.synthetic_weights(dims, seed=3)
    g = torch.Generator().manual_seed(11)
    n_clips = 300  # the "dataset"; the step draws 256 of them
    xa = torch.randn(n_clips, dims.n_audio_ctx, dims.n_audio_state, generator=g) * 0.5
    rng = np.random.default_rng(5)
    rows = []
    for i in range(n_clips):
        n = int(rng.integers(2, 9))
        rows.append([1, 2] + rng.integers(3, 80, size=n).tolist() + [TINY_EOT] * (11 - 2 - n))
    tokens = torch.tensor(rows, dtype=torch.int64)
    draw = [int(i) for i in np.random.default_rng(0).choice(n_clips, 256, replace=False)]
    return R, dims, W, xa, tokens, draw


def _tiny_flat_layout(W):
    names = [k for k in W if k.startswith("decoder.")]
    offs, total = {}, 0
    for k in names:
        offs[k] = total
        total += W[k].numel()
    # segments in the order the trainer's backward finishes them: [final ln] <- block 1 <- block 0 <- [embeddings]
    first_block = [min(offs[k] for k in names if k.startswith(f"decoder.blocks.{l}.")) for l in range(2)]
    tail = min(offs[k] for k in names if k.startswith("decoder.ln."))
    assert first_block[0] < first_block[1] < tail  # parameter_names() order: embeddings, blocks, final ln
    segs = [(tail, total), (first_block[1], tail), (first_block[0], first_block[1]), (0, first_block[0])]
    return names, offs, total, segs


def _dp8_worker(rank, world, port, q):
    import torch.distributed as dist
    from whisper_ipa_amd import parallel as P

    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_WORLD_SIZE"] = str(world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        R, dims, W, xa, tokens, draw = _tiny_dp_problem()
        per = P.require_even_shards(len(draw), world)                     # 256 -> 8 x 32
        mine = P.shard_indices(draw, world, rank)                         # slice r of the ONE shared draw
        width = P.agree_on_step(int(tokens[mine].shape[1]) - (rank % 3))  # ranks may see narrower batches: the MAX is used
        names, offs, total, segs = _tiny_flat_layout(W)
        leaves = {k: W[k].detach().clone().requires_grad_(True) for k in names}
        Wl = dict(W)
        Wl.update(leaves)
        tok = tokens[mine]
        logits = R.decoder_forward(Wl, dims, tok[:, :-1], xa[mine])
        tgt = tok[:, 1:]
        mask = R.loss_mask(tgt, TINY_EOT).reshape(-1)
        ce = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), tgt.reshape(-1), reduction="none")
        local_sum = torch.where(mask, ce, torch.zeros_like(ce)).sum()
        # exchange 1: (sum CE, valid count) BEFORE the backward, so every rank divides by the GLOBAL count
        g_sum, g_cnt = P.allreduce_loss_stats(local_sum.detach(), mask.sum())
        grads = torch.autograd.grad(local_sum / torch.clamp(g_cnt, min=1), [leaves[k] for k in names])
        flat = torch.empty(total)
        for k, gk in zip(names, grads):
            flat[offs[k]: offs[k] + gk.numel()] = gk.reshape(-1)
        # exchange 2: finished segments of the flat buffer, asynchronously, in the backward's order
        red = P.SegmentReducer(flat)
        for lo, hi in segs:
            red.reduce(lo, hi)
        joined = red.wait()
        fail_seen = P.agree_on_step(11, failed=(rank == 5))  # one rank cannot build its batch: EVERY rank must see -1
        # the clip runs AFTER the exchange, on the reduced gradients, over the tensors the trainer's seg_clip flags mark
        # (training.clip_reaches: what wipa_clip_adamw is handed); the arithmetic below is the kernel's, per tensor
        from whisper_ipa_amd.training import clip_reaches

        after = {}
        if rank in (0, 7):
            for scope in ("reference", "all"):
                out = flat.clone()
                for k in names:
                    gk = out[offs[k]: offs[k] + W[k].numel()]
                    if clip_reaches(k, scope):
                        gk *= torch.clamp(1.0 / (torch.sqrt(torch.sum(gk * gk)) + 1e-6), max=1.0)
                after[scope] = out.numpy().tobytes()
        q.put((rank, per, mine, width, float(g_sum / g_cnt), int(g_cnt), joined, fail_seen, P.host_threads_per_rank(cores=64),
               flat.numpy().tobytes() if rank in (0, 7) else None, after))
    finally:
        dist.destroy_process_group()


def test_dp_world8_global_batch_256_gradients_equal_single_process():
    """VERDICT r3 next #7(a) / SURVEY 8(e) / BASELINE configs[2]: world 8 has never run anywhere.  Eight gloo ranks train ONE
    step of a tiny decoder on a global batch of 256 = 8 x 32 (require_even_shards, shard_indices of one shared draw,
    agree_on_step, allreduce_loss_stats before the backward, SegmentReducer over the flat gradient buffer in the backward's
    segment order): loss and the reduced gradients on every rank equal the single-process loss / autograd over the 256 clips
    (reference scripts/train_whisper_ipa.py:260-261,287-303,548).  Round 5: the clip that follows the exchange is checked
    against the oracle's TREE-WALKING clip_grad_dict (:287-303 as written: the decoder.blocks list passes through) on the
    single-process gradients, in both scopes; un-clipped norms above 1 exist among the block tensors and outside them."""
    import torch.multiprocessing as mp

    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 1000
    procs = [ctx.Process(target=_dp8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    R, dims, W, xa, tokens, draw = _tiny_dp_problem()
    names, offs, total, segs = _tiny_flat_layout(W)
    leaves = {k: W[k].detach().clone().requires_grad_(True) for k in names}
    Wl = dict(W)
    Wl.update(leaves)
    loss = R.loss_from_features(Wl, dims, xa[draw], tokens[draw], TINY_EOT)
    grads = torch.autograd.grad(loss, [leaves[k] for k in names])
    want = torch.cat([g.reshape(-1) for g in grads])
    assert [r[0] for r in res] == list(range(world))
    assert sum((r[2] for r in res), []) == draw and all(r[1] == 32 and len(r[2]) == 32 for r in res)  # the shared draw, in rank order
    n_valid = int(R.loss_mask(tokens[draw][:, 1:], TINY_EOT).sum())
    for r in res:
        assert r[3] == tokens.shape[1]  # agree_on_step: the widest batch
        assert abs(r[4] - float(loss.detach())) < 1e-5 and r[5] == n_valid  # the GLOBAL loss on every rank
        assert r[6] == len(segs) and r[7] == -1
        assert r[8] == 8  # 64 usable cores shared by LOCAL_WORLD_SIZE = 8 ranks
    for r in (res[0], res[7]):
        got = torch.from_numpy(np.frombuffer(r[9], dtype=np.float32).copy())
        assert got.shape == want.shape
        assert float((got - want).abs().max() / want.abs().max()) < 2e-5  # same gradients as ONE process over 256 clips
    assert float(want.abs().max()) > 1e-4
    g_single = {k: g.detach() for k, g in zip(names, grads)}
    from parity_util import clipped_norm, norm64

    over = [k for k in names if norm64(g_single[k]) > 1.0]
    assert any(".blocks." in k for k in over) and any(".blocks." not in k for k in over), over
    for scope in ("reference", "all"):
        ref = R.clip_gradients(g_single, 1.0, scope)  # scope "reference": unflatten -> clip_grad_dict (dicts only) -> flatten
        want_c = torch.cat([ref[k].reshape(-1) for k in names])
        for r in (res[0], res[7]):
            got = torch.from_numpy(np.frombuffer(r[10][scope], dtype=np.float32).copy())
            assert float((got - want_c).abs().max() / want_c.abs().max()) < 2e-5, scope
        for k in over:
            n_after = norm64(ref[k])
            if ".blocks." in k and scope == "reference":
                assert n_after > 1.0 and torch.equal(ref[k], g_single[k])  # the list is handed back untouched
            else:
                assert abs(n_after - clipped_norm(norm64(g_single[k]))) < 1e-6, (k, n_after)  # train_whisper_ipa.py:295-298
    blk = [k for k in over if ".blocks." in k][0]
    assert not torch.equal(R.clip_gradients(g_single, 1.0, "reference")[blk], R.clip_gradients(g_single, 1.0, "all")[blk])


class _FakeValModel:
    """decode() returns the clip's index spelled out, so the gathered hypotheses reveal which rank decoded what"""

    def __init__(self):
        self.decoded = []

    def eval(self):
        return self

    def train(self):
        return self

    def decode(self, mel, options):
        from types import SimpleNamespace

        assert options.language is None and options.fp16 is False
        ids = [int(v) for v in mel[:, 0, 0].tolist()]
        self.decoded += ids
        return [SimpleNamespace(text=" " + "ab"[i % 2] * (1 + i % 3) + " ") for i in ids]


class _FakeValData:
    def get_batch(self, indices):
        if 9 in indices:
            raise OSError("unreadable clip")  # the reference keeps validating (train_whisper_ipa.py:393-396)
        mel = torch.zeros(len(indices), 2, 2)
        mel[:, 0, 0] = torch.tensor(indices, dtype=torch.float32)
        return {"mel_features": mel, "tokens": torch.tensor([[50258, 97 + i % 2, 50257] for i in indices])}


class _FakeValTok:
    def decode(self, ids):
        return "".join("<|sot|>" if t == 50258 else "<|eot|>" if t == 50257 else chr(t) for t in ids)


def _fake_transcribe_batches(model, batches, options, passes_in_flight=4, **kw):
    """pipeline.transcribe_batches' contract as validate() uses it (lazy iteration, one result per batch in input order with
    .index and .texts), on the fake model: no GPU in this test"""
    from types import SimpleNamespace

    assert passes_in_flight >= 1
    for i, mel in enumerate(batches):
        yield SimpleNamespace(index=i, texts=[r.text for r in model.decode(mel, options)])


def _validate_worker(rank, world, port, q):
    import torch.distributed as dist

    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import train_whisper_ipa as T

    T.transcribe_batches = _fake_transcribe_batches

    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _FakeValModel()
        m = T.validate(model, _FakeValData(), _FakeValTok(), num_samples=22)
        q.put((rank, m["per"], m["pfer"], m["num_samples"], sorted(model.decoded)))
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_validate_is_sharded_over_dp_ranks_gloo_world2():
    """VERDICT r2 missing #5: validate() under data parallelism is collective -- the 4-clip validation batches are dealt
    round-robin to the ranks, the (reference, hypothesis) pairs are gathered, every rank returns the metrics a single
    process computes on the whole list (reference scripts/train_whisper_ipa.py:314-407, :568-588)."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + os.getpid() % 1000
    procs = [ctx.Process(target=_validate_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    q1 = ctx.Queue()
    _validate_worker(0, 1, 0, q1)
    single = q1.get(timeout=10)
    assert single[3] == 18 and single[4] == [i for i in range(22) if not 8 <= i < 12]  # the batch with the unreadable clip is skipped
    for r in res:
        assert r[1:4] == single[1:4]  # same PER / PFER / sample count on every rank as in one process
    assert res[0][4] == [0, 1, 2, 3, 16, 17, 18, 19] and res[1][4] == [4, 5, 6, 7, 12, 13, 14, 15, 20, 21]  # rank 0's batch 2 failed
    assert single[1] > 0  # the fake hypotheses are not all correct: the metric is not vacuous


def test_shard_bounds_cover_everything():
    from whisper_ipa_amd.parallel import SegmentReducer, agree_on_step, shard_bounds

    for n in (0, 1, 7, 64, 65):
        for w in (1, 2, 4, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
    # a DP training batch must split evenly (the default --batch-size 12 on 8 ranks would leave ranks 6, 7 empty)
    from whisper_ipa_amd.parallel import require_even_shards

    assert require_even_shards(256, 8) == 32 and require_even_shards(12, 1) == 12
    for bad in ((12, 8), (4, 8), (0, 2)):
        with pytest.raises(ValueError):
            require_even_shards(*bad)
    # single process: the DP helpers are no-ops with the same return contract
    flat = torch.arange(10.0)
    red = SegmentReducer(flat)
    red.reduce(0, 10)
    assert red.wait() == 0 and torch.equal(flat, torch.arange(10.0))
    assert agree_on_step(12) == 12 and agree_on_step(12, failed=True) == -1


def test_evaluate_ipa_tokenisation_known_answers():
    """the nine assertions of the reference's scripts/evaluate_ipa.py:449-457 + PER examples."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import evaluate_ipa as ev

    assert ev.tokenize_ipa("n̩æp") == ["n̩", "æ", "p"]
    assert ev.tokenize_ipa("ɾ̃æ") == ["ɾ̃", "æ"]
    assert ev.tokenize_ipa("ə̥tʃ") == ["ə̥", "t", "ʃ"]
    assert ev.tokenize_ipa("tʃ") == ["t", "ʃ"]
    assert ev.tokenize_ipa("ŋ̍") == ["ŋ̍"]
    assert ev.tokenize_ipa("kæt") == ["k", "æ", "t"]
    assert ev.tokenize_ipa("m̩") == ["m̩"] and ev.tokenize_ipa("l̩") == ["l̩"] and ev.tokenize_ipa("") == []
    assert ev.tokenize_ipa("tʰ a") == ["tʰ", "a"]
    assert ev.phone_error_rate("kæt", "kæt") == 0.0
    assert abs(ev.phone_error_rate("kæt", "kat") - 100.0 / 3) < 1e-9
    assert ev.phone_error_rate("", "") == 0.0 and ev.phone_error_rate("", "a") == 100.0
    assert ev.edit_distance("kitten", "sitting") == 3
    assert ev.normalize_ipa_for_comparison("g a") == "ɡa"
    ev.set_feature_table(None)
    m = ev.evaluate_batch(["kæt", "dɔɡ"], ["kæt", "dɔ"])
    assert m["num_samples"] == 2 and abs(m["per"] - (0 + 100.0 / 3) / 2) < 1e-9
    assert m["per_scores"] == [0.0, 1 / 3 * 100.0] and len(m["pfer_scores"]) == 2  # per-sample scores (reference :370-378)
    # the RAW strings are scored: Latin g vs IPA ɡ is an error here, as in the reference's evaluate_batch (:363-368)
    assert ev.evaluate_batch(["dɔɡ"], ["dɔg"])["per"] > 0
    if m["pfer_is_per_fallback"]:  # no panphon / WIPA_PANPHON_CSV in this image
        assert m["pfer"] == m["per"]
        with pytest.raises(RuntimeError):
            ev.phone_feature_error_rate("kæt", "kat")


class _ToyFeatures:
    """A 24-feature table for five phones (hand-made, NOT panphon's values) to drive the PFER dynamic programs."""

    def __init__(self):
        z = [0] * 24
        self.v = {"p": [1] * 24, "b": [1] * 23 + [-1], "m": [1] * 12 + [-1] * 12, "a": [-1] * 24, "x": z}
        self.v["b̥"] = [1] * 22 + [0, -1]  # a diacritic variant: differs from b in ONE feature (+1 vs 0)

    def word_to_vector_list(self, word, numeric=True):
        return [self.v[word]] if word in self.v else []

    def ipa_segs(self, text):
        return [c for c in text if c != "̥"]  # like panphon on some inputs: DROPS a character -> Unicode fallback


def test_pfer_hamming_and_cosine_dynamic_programs():
    """reference evaluate_ipa.py:139-213 (Hamming: substitution = #features that differ / 24, a 0-vs-(+1) difference counts 1,
    not 1/2) and :216-287 (cosine: every operation costs 1 - cos when the vectors differ)."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import evaluate_ipa as ev

    ev.set_feature_table(_ToyFeatures())
    try:
        calc = ev.get_pfer_calculator()
        assert calc.feature_distance("p", "p") == 0.0
        assert abs(calc.feature_distance("p", "b") - 1 / 24) < 1e-12
        assert abs(calc.feature_distance("b", "b̥") - 1 / 24) < 1e-12  # +1 vs 0 is ONE mismatch (not 0.5)
        assert abs(calc.feature_distance("p", "m") - 12 / 24) < 1e-12
        assert abs(calc.feature_distance("p", "q") - 1.0) < 1e-12     # unknown phone -> zero vector: all 24 differ
        # segmentation: the table's ipa_segs is used when it keeps every character, the Unicode rule otherwise
        assert ev.tokenize_ipa("pam") == ["p", "a", "m"] and ev.tokenize_ipa("b̥a") == ["b̥", "a"]
        # one substitution p->b in three phones
        assert abs(ev.phone_feature_error_rate("pam", "bam") - (1 / 24) / 3 * 100) < 1e-9
        # deletion costs 1, cheaper than nothing else
        assert abs(ev.phone_feature_error_rate("pam", "pm") - 1 / 3 * 100) < 1e-9
        # substitution (12/24) beats delete+insert (2)
        assert abs(ev.phone_feature_error_rate("p", "m") - 50.0) < 1e-9
        assert ev.phone_feature_error_rate("", "") == 0.0 and ev.phone_feature_error_rate("", "p") == 100.0
        # cosine: p vs a are opposite vectors (1 - (-1) = 2); p vs b: 1 - 22/24
        assert abs(ev.phone_feature_error_rate_cosine("p", "a") - 200.0) < 1e-9
        assert abs(ev.phone_feature_error_rate_cosine("pa", "ba") - (1 - 22 / 24) / 2 * 100) < 1e-9
        assert ev.phone_feature_error_rate_cosine("pam", "pam") == 0.0
        # equal vectors continue the diagonal even when the strings differ; zero vectors use the 0.001 guard
        assert abs(ev.phone_feature_error_rate_cosine("x", "p") - 100.0) < 1e-9
        m = ev.evaluate_batch(["pam", "p"], ["bam", "m"])
        assert m["pfer_is_per_fallback"] is False and abs(m["pfer"] - ((1 / 24) / 3 * 100 + 50.0) / 2) < 1e-9
        assert abs(m["per"] - (100 / 3 + 100.0) / 2) < 1e-9 and m["pfer"] < m["per"]
    finally:
        ev.set_feature_table(None)


def test_csv_feature_table_reads_the_panphon_layout(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import evaluate_ipa as ev

    names = [f"f{i}" for i in range(24)]
    rows = [["ipa"] + names, ["p"] + ["+"] * 24, ["b"] + ["+"] * 23 + ["-"], ["a"] + ["-"] * 12 + ["0"] * 12]
    p = tmp_path / "ipa_all.csv"
    p.write_text("\n".join(",".join(r) for r in rows) + "\n", encoding="utf-8")
    ft = ev.CsvFeatureTable(str(p))
    assert ft.word_to_vector_list("b")[0][-1] == -1 and ft.word_to_vector_list("a")[0][-1] == 0 and ft.word_to_vector_list("z") == []
    ev.set_feature_table(ft)
    try:
        assert abs(ev.phone_feature_error_rate("pa", "ba") - (1 / 24) / 2 * 100) < 1e-9
        # a diacritic / modifier without a row of its own (ejective p', aspirated b, nasalised a) is scored with its
        # BASE character's features, not with the all-zero vector, and the result says which phones that happened to; a symbol
        # with no row at all keeps the reference's zero vector (evaluate_ipa.py:130-137) and is reported too
        assert ft.word_to_vector_list("p\u02bc") == ft.word_to_vector_list("p")
        assert ft.word_to_vector_list("a\u0303") == ft.word_to_vector_list("a")
        m = ev.evaluate_batch(["p\u02bca", "bz"], ["pa", "b\u02b0a\u0303"])
        assert m["pfer_base_fallback_phones"].keys() == {"p\u02bc", "b\u02b0", "a\u0303"}
        assert m["pfer_unknown_phones"].keys() == {"z"}
        # p' vs p: same base features -> substitution cost 0 although the strings differ; PER counts it as an error
        assert m["pfer_scores"][0] == 0.0 and m["per_scores"][0] == 50.0
    finally:
        ev.set_feature_table(None)


def test_bench_synthetic_inputs_equal_the_oracles():
    """bench.py restates the oracle's seeded generators (the product path may not import oracle/): same clips sample for
    sample and same weights tensor for tensor, so `parity_vs_cpu` and the full-depth GPU tests compare like with like."""
    import bench
    from oracle import whisper_ref as R

    a = bench.synthetic_audio(3, 2)
    assert a.dtype == np.float32 and (a[0] == R.synthetic_clip(3)).all() and (a[1] == R.synthetic_clip(4)).all()
    dims, W = bench.synthetic_weights_small(0, "tiny")
    W2 = R.synthetic_weights(R.DIMS["tiny"], seed=0)
    assert dims.__dict__ == R.DIMS["tiny"].__dict__
    assert set(W) == set(W2) and all(torch.equal(W[k], W2[k]) for k in W)
    # the "peaky" preset (a pointer to a seeded token sequence in the decoder's positional table) is the same on both sides
    _, Wp = bench.synthetic_weights_small(0, "tiny", preset="peaky")
    Wp2 = R.synthetic_weights(R.DIMS["tiny"], seed=0, preset="peaky")
    assert all(torch.equal(Wp[k], Wp2[k]) for k in Wp)
    changed = [k for k in W if not torch.equal(W[k], Wp[k])]
    assert changed == ["decoder.positional_embedding"]


def test_bench_launcher_refuses_more_ranks_than_gpus(monkeypatch, capsys):
    """`bench.py --gpus N` without WORLD_SIZE is a launcher; it never touches the GPU itself (the device count comes from the
    KFD topology in sysfs, not from HIP) and refuses (exit 2) when the node has fewer than N devices instead of silently
    running one rank (round-1 behaviour)."""
    import bench

    monkeypatch.delenv("WIPA_BENCH_SHARE_GPU", raising=False)
    monkeypatch.setattr(bench, "visible_gpu_count", lambda: 1)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: (_ for _ in ()).throw(AssertionError("the launcher must not ask HIP")))
    assert bench.launch_ranks(8, ["--gpus", "8"]) == 2
    assert "only 1 GPU" in capsys.readouterr().err
    monkeypatch.undo()
    n = bench.visible_gpu_count()
    assert n is None or n >= 1  # no /sys/class/kfd in this container: unknown -> the ranks' own device check decides
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    assert bench.visible_gpu_count() in (None, 1)


def test_bench_launcher_terminates_the_other_ranks_when_one_dies(monkeypatch, tmp_path, capsys):
    """a rank that dies before the rendezvous must not leave the others waiting with the GPU held.  The launcher
    watches every child: here rank 1 exits with 3 at once while rank 0 would sleep for a minute -- the launcher returns 3
    within seconds and rank 0 is gone."""
    import time

    import bench

    script = tmp_path / "fake_rank.py"
    script.write_text("import os, sys, time\n"
                      "if os.environ['RANK'] == '1':\n    sys.exit(3)\n"
                      "open(os.environ['FAKE_PID_FILE'], 'w').write(str(os.getpid()))\ntime.sleep(60)\n")
    monkeypatch.setattr(bench, "__file__", str(script))
    monkeypatch.setattr(bench, "visible_gpu_count", lambda: None)
    monkeypatch.setenv("FAKE_PID_FILE", str(tmp_path / "pid"))
    t0 = time.time()
    rc = bench.launch_ranks(2, [])
    assert rc == 3 and time.time() - t0 < 30
    assert "terminating the other ranks" in capsys.readouterr().err
    pid = int((tmp_path / "pid").read_text()) if (tmp_path / "pid").exists() else None
    if pid is not None:
        import os
        assert not os.path.exists(f"/proc/{pid}") or open(f"/proc/{pid}/stat").read().split()[2] == "Z"


# The ONLY reference-held golden vectors on the hot path: Whisper multilingual BPE ids of IPA strings
# (/root/reference/WHISPER_IPA_RESEARCH_STANDALONE.md:280-305 "IPA tokenisation examples" and :498-505), and the special
# ids of the <= large-v2 multilingual vocabulary (:333-338).
REFERENCE_KNOWN_ANSWER_IDS = {
    "ə": [7250], "θ": [9440], "æ": [7303], "ʃ": [133, 103], "ɛ̃": [133, 249, 136, 225], "əː": [7250, 135, 238],
    "tʰ": [83, 134, 108], "n̩": [77, 136, 102], "p": [79], "t": [83], "a": [64],
}


def test_special_token_ids_match_the_reference_table():
    """WHISPER_IPA_RESEARCH_STANDALONE.md:333-338: eot 50257, sot 50258, <|en|> 50259, transcribe 50359, notimestamps 50363 --
    these do not depend on the rank table, so they are pinned in byte-fallback mode too."""
    from whisper_ipa_amd.tokenizer import get_tokenizer

    tok = get_tokenizer(True)
    assert (tok.eot, tok.sot, tok.to_language_token("en"), tok.transcribe, tok.no_timestamps) == (50257, 50258, 50259, 50359, 50363)
    assert tuple(tok.sot_sequence_including_notimestamps) == (50258, 50259, 50359, 50363)


@pytest.mark.skipif(not os.environ.get("WIPA_TIKTOKEN"), reason="Whisper vocabulary (multilingual.tiktoken) is not in this image: "
                    "set WIPA_TIKTOKEN to arm the reference's known-answer ids")
def test_tokenizer_reference_known_answer_ids():
    from whisper_ipa_amd.tokenizer import get_tokenizer

    tok = get_tokenizer(True)
    assert not tok.byte_fallback and len(tok.ranks) == 50257
    for text, ids in REFERENCE_KNOWN_ANSWER_IDS.items():
        assert tok.encode(text) == ids, (text, tok.encode(text), ids)
        assert tok.decode(ids) == text


def test_byte_level_known_answers_hold_without_the_vocabulary():
    """The byte -> rank permutation of the GPT-2 vocabulary family is fixed, so every reference known answer that involves no
    MERGE is checkable here, without the rank table: p, t, a, ɪ, ɛ, ɛ̃, tʰ, n̩ in full and the unmerged tail of əː.
    (The reference's table also lists 'ʃ' -> [133, 103]; that is ɪ's encoding (bytes C9 AA) -- ʃ is CA 83 and cannot share
    it, so it is recorded as a documentation slip of the reference, not asserted.)"""
    from whisper_ipa_amd.tokenizer import Tokenizer, gpt2_byte_ranks

    tok = Tokenizer(gpt2_byte_ranks(), byte_fallback=True)
    unmerged = {"p": [79], "t": [83], "a": [64], "ɪ": [133, 103], "ɛ": [133, 249], "ɛ̃": [133, 249, 136, 225],
                "tʰ": [83, 134, 108], "n̩": [77, 136, 102]}
    for text, ids in unmerged.items():
        assert tok.encode(text) == ids, (text, tok.encode(text), ids)
        assert tok.decode(ids) == text
    assert tok.encode("əː")[-2:] == [135, 238]          # 'əː' -> [7250, 135, 238]: the length mark is unmerged
    assert tok.encode("ə") == [133, 247]                # the English-only column of the reference's table (:297-302): no merge
    assert tok.encode("θ") == [138, 116]                # ditto
    assert tok.encode(" ") == [220]                     # SuppressBlank's token
    assert tok.encode("ʃ") == [134, 225] != tok.encode("ɪ")
    assert sorted(gpt2_byte_ranks().values()) == list(range(256))


def test_byte_fallback_is_refused_for_real_weight_entry_points(monkeypatch, tmp_path, capsys):
    """fine-tuning / scoring a pretrained checkpoint on byte-fallback ids must not happen silently."""
    from whisper_ipa_amd import tokenizer as T

    monkeypatch.delenv("WIPA_ALLOW_BYTE_FALLBACK", raising=False)
    tok = T.get_tokenizer(True)
    if not tok.byte_fallback:
        pytest.skip("real vocabulary loaded")
    with pytest.raises(T.VocabularyError):
        T.require_real_vocabulary(tok, False, "test")
    assert T.require_real_vocabulary(tok, True, "test") is tok
    monkeypatch.setenv("WIPA_ALLOW_BYTE_FALLBACK", "1")
    assert T.require_real_vocabulary(tok, False, "test") is tok
    monkeypatch.delenv("WIPA_ALLOW_BYTE_FALLBACK")
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import ipa_data_loader as D

    p = tmp_path / "d.json"
    p.write_text("[]")
    with pytest.raises(T.VocabularyError):
        D.create_data_loader(str(p))
    assert len(D.create_data_loader(str(p), allow_byte_fallback=True)) == 0
    with pytest.raises(FileNotFoundError):
        T.get_tokenizer(True, vocab_path=str(tmp_path / "missing.tiktoken"))


def _e4m3fn_table():
    """All 256 OCP e4m3fn codes decoded from the format definition: 1 sign, 4 exponent (bias 7), 3 mantissa bits; exponent 0 is
    subnormal (2^-6 * m/8); S.1111.111 is NaN, there are no infinities (largest finite 448)."""
    out = []
    for c in range(256):
        s, e, m = c >> 7, (c >> 3) & 15, c & 7
        if e == 15 and m == 7:
            v = float("nan")
        elif e == 0:
            v = 2.0 ** -6 * (m / 8.0)
        else:
            v = 2.0 ** (e - 7) * (1 + m / 8.0)
        out.append(-v if s else v)
    return out


def test_fp8_e4m3_quantiser():
    """whisper.quantize_fp8_e4m3 (BASELINE.json configs[4]): OCP e4m3fn codes, per-row power-of-two scales, round to nearest,
    the dequantised value exact in bf16."""
    from whisper_ipa_amd.whisper import FP8_MAX, dequantize_fp8_e4m3, quantize_fp8_e4m3

    table = torch.tensor(_e4m3fn_table(), dtype=torch.float64)
    assert table[0x7E] == FP8_MAX == 448.0 and table[0x08] == 2.0 ** -6 and table[0x01] == 2.0 ** -9
    g = torch.Generator().manual_seed(0)
    W = torch.randn(37, 3, 64, generator=g) * torch.logspace(-4, 2, 37)[:, None, None]  # rows of very different magnitude
    W[5] = 0
    codes, scale = quantize_fp8_e4m3(W)
    assert codes.dtype == torch.uint8 and tuple(codes.shape) == (37, 192) and tuple(scale.shape) == (37,)
    assert torch.equal(torch.exp2(torch.round(torch.log2(scale))), scale)  # powers of two
    assert not ((codes & 0x7F) == 0x7F).any()  # no NaN codes
    dq = dequantize_fp8_e4m3(codes, scale)
    assert torch.equal(dq.double(), table[codes.long()] * scale[:, None].double())  # the format definition, code by code
    assert torch.equal(dq.to(torch.bfloat16).float(), dq)  # exactly representable in bf16
    W2 = W.reshape(37, -1)
    amax = W2.abs().amax(1)
    nz = amax > 0
    assert ((amax[nz] / scale[nz]) <= 448).all() and ((amax[nz] / scale[nz]) > 224).all()  # the smallest such power of two
    # round to nearest: no other code of the row's grid is closer
    err = (dq - W2).abs()
    grid = table[:0x7F].float()  # non-negative finite values
    for r in (0, 11, 36):
        cand = grid[None, :] * scale[r]
        best = (cand - W2[r].abs()[:, None]).abs().min(1).values
        assert torch.allclose(err[r], best, rtol=0, atol=1e-12 * float(scale[r]) + 1e-30)
    big = W2.abs() > 2.0 ** -6 * scale[:, None] * 8  # normal range: relative error <= 2^-4
    assert (err[big] / W2.abs()[big]).max() <= 2.0 ** -4
    assert (dq[5] == 0).all()


def test_native_bpe_matches_the_python_merge_loop(tmp_path):
    """csrc/text_host.cpp (wipa_bpe_*) against the pure-Python statement of tiktoken's lowest-rank-first merging, on a
    synthetic table with competing merges, on every IPA string of the reference's test split (byte-level table) and on
    random byte strings; decode round trips; the batch builder against the list arithmetic of ipa_data_loader.py:102-131."""
    import ctypes as C

    from whisper_ipa_amd import _lib
    from whisper_ipa_amd.tokenizer import Tokenizer, _NativeBPE, _bpe, gpt2_byte_ranks

    ranks = dict(gpt2_byte_ranks())
    for i, t in enumerate([b"ab", b"bc", b"abc", b"cd", b"abcd", b" a", "ə".encode(), "əː".encode(), b"aa", b"aaa", b"aaaa", b"\xc9\x99\xcb"]):
        ranks[t] = 256 + i
    nat = _NativeBPE(ranks)
    rng = np.random.default_rng(0)
    cases = [b"abcd", b"abcabcd", b"aaaaaaa", b"bcd", "əːə".encode(), b" ab", b"x", b"", bytes(range(256))]
    cases += [bytes(rng.choice(list(b"abcd \xc9\x99\xcb\x90"), size=int(n))) for n in rng.integers(1, 40, size=200)]
    for piece in cases:
        want = _bpe(ranks, piece) if piece else []
        got = nat.encode_piece(piece)
        assert got == want, (piece, got, want)
        assert nat.decode_bytes(got) == piece
    assert nat.decode_bytes([5, 99999]) is None  # unknown id: left to the caller (special tokens)
    tok = Tokenizer(gpt2_byte_ranks(), byte_fallback=True)
    p = "/root/reference/data/v2_filtered/combined_test_ipa.json"
    texts = [e["ipa_transcription"] for e in json.load(open(p))[:300]] if os.path.exists(p) else ["kæt n̩ tʰ əː", "ɛ̃ ʃ"]
    for s in texts:
        ids = tok.encode(s)
        assert ids == [i for piece in tok._pat.findall(s) for i in _bpe(tok.ranks, piece.encode("utf-8"))]
        assert tok.decode(ids) == s
    # batch builder
    L = _lib.lib()
    rows = [[5, 6, 7], [], [9] * 11, [1]]
    prefix, eot = [50258, 50259, 50359, 50363], 50257
    flat = [i for r in rows for i in r]
    out = torch.full((4, 20), -1, dtype=torch.int32)
    w = L.wipa_build_token_batch((C.c_int32 * len(flat))(*flat), (C.c_int32 * 4)(*[len(r) for r in rows]), 4, (C.c_int32 * 4)(*prefix), 4, eot,
                                 out.data_ptr(), 20)
    assert w == 4 + 11 + 1
    want = [prefix + r + [eot] * (w - 4 - len(r)) for r in rows]
    assert out[:, :w].tolist() == want and (out[:, w:] == -1).all()
    assert L.wipa_build_token_batch((C.c_int32 * len(flat))(*flat), (C.c_int32 * 4)(*[len(r) for r in rows]), 4, (C.c_int32 * 4)(*prefix), 4, eot,
                                    out.data_ptr(), 10) < 0  # rows do not fit


def test_hw_queue_request_and_batch_prefetch(monkeypatch):
    """pipeline's host-side plumbing without a GPU: the hardware-queue request never overrides the user's GPU_MAX_HW_QUEUES,
    sets 8 when nothing has initialised the GPU, reports the ROCm default when it came too late; the prefetching iterator
    keeps order and surfaces the producer's exception in the consumer."""
    import whisper_ipa_amd.runtime as RT
    from whisper_ipa_amd.pipeline import _prefetched

    assert os.environ.get("GPU_MAX_HW_QUEUES") is not None  # importing the package asked for it (or the user had set it)
    monkeypatch.setitem(RT._hwq, "requested_in_time", None)
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "6")
    assert RT.request_hw_queues() == 6 and RT.hw_queues() == 6  # the user's value stands
    monkeypatch.delenv("GPU_MAX_HW_QUEUES")
    monkeypatch.setitem(RT._hwq, "requested_in_time", None)
    assert RT.request_hw_queues() == 8 and os.environ["GPU_MAX_HW_QUEUES"] == "8" and RT.hw_queues() == 8
    monkeypatch.delenv("GPU_MAX_HW_QUEUES")
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: True)
    assert RT.request_hw_queues() == RT.ROCM_DEFAULT_HW_QUEUES and "GPU_MAX_HW_QUEUES" not in os.environ  # too late: not pretended
    assert RT.hw_queues() == RT.ROCM_DEFAULT_HW_QUEUES
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")  # set after the runtime started: not in effect
    assert RT.hw_queues() == RT.ROCM_DEFAULT_HW_QUEUES

    assert list(_prefetched(iter(range(7)), 2)) == list(range(7))

    def bad():
        yield 1
        raise OSError("unreadable")

    it = _prefetched(bad(), 1)
    assert next(it) == 1
    with pytest.raises(OSError):
        next(it)


def test_bench_optional_legs_do_not_cost_the_line(capsys):
    """bench.py: an exception in an OPTIONAL leg (evaluate-style, MFMA roofline, fine-tune, other_configs -- e.g. out of memory in the
    large-v3 leg) is recorded under `leg_errors` with its traceback on stderr; the headline and the other legs survive.  The id checks,
    `roofline`, `decode_step` and `cpu_baseline` are not wrapped: they stay hard failures."""
    import bench

    out = {}
    assert bench.optional_leg(out, "fine", lambda: {"v": 1}) == {"v": 1} and "leg_errors" not in out
    assert bench.optional_leg(out, "configs[4]", lambda: (_ for _ in ()).throw(MemoryError("HIP out of memory"))) is None
    assert "MemoryError" in out["leg_errors"]["configs[4]"]
    err = capsys.readouterr().err
    assert "OPTIONAL LEG 'configs[4]' FAILED" in err and "Traceback" in err
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    for hard in ('out["roofline"] = roofline_cross_attn(model, B)', 'out["decode_step"] = decode_step_roofline(model, B)',
                 'out["cpu_baseline"], ref, xa_ref = cpu_baseline(8)', 'assert (single == tokens).all()'):
        assert hard in main, hard  # not behind optional_leg
