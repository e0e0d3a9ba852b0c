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
    assert t[0].tolist() == [50258, 50259, 50359, 50363, 97, 98, 50257, 50257, 50257, 50257]
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
        torch.manual_seed(0)
        full = {f"g{i}": torch.randn(300 + i) for i in range(5)}
        part = {k: v * (0.25 if rank == 0 else 0.75) for k, v in full.items()}
        P.allreduce_grads(part, bucket_bytes=2000)
        ok = all(torch.allclose(part[k], full[k], atol=1e-6) for k in full)
        q.put((rank, gathered == rows_all, [int(i) for i in mine], float(s), float(n), ok))
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


def test_shard_bounds_cover_everything():
    from whisper_ipa_amd.parallel import bucketed, shard_bounds

    for n in (0, 1, 7, 64, 65):
        for w in (1, 2, 4, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
    named = {f"t{i}": torch.zeros(1000) for i in range(10)}
    buckets = list(bucketed(named, bucket_bytes=12000))
    assert [n for b in buckets for n in b] == list(named) and max(len(b) for b in buckets) == 3


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
    m = ev.evaluate_batch(["kæt", "dɔɡ"], ["kæt", "dɔ"])
    assert m["num_samples"] == 2 and abs(m["per"] - (0 + 100.0 / 3) / 2) < 1e-9 and "pfer" in m
