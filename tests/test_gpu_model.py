"""GPU parity tests, path level: log-mel -> encoder -> (teacher-forced | KV-cached greedy)
decoder through the C ABI, against the CPU oracle on the same seeded inputs and against the
committed golden fixtures.  ``pytest -m gpu`` on an MI355X."""
import os

import numpy as np
import pytest
import torch

from oracle import whisper_ref as R
from parity_util import check_low_precision_decode

pytestmark = pytest.mark.gpu

MICRO = R.ModelDimensions(80, 1500, 128, 2, 2, 51865, 448, 128, 2, 2)
SMALL2 = R.ModelDimensions(80, 1500, 768, 12, 2, 51865, 448, 768, 12, 2)  # whisper-small width, 2+2 layers


def _model(dims_o, W, dtype, f32_split=False, cross_attention="auto"):
    from whisper_ipa_amd.whisper import ModelDimensions, Whisper

    m = Whisper(ModelDimensions(**dims_o.__dict__), dtype=dtype, f32_split=f32_split, cross_attention=cross_attention)
    m.load_weights(W)
    return m


@pytest.fixture(scope="module")
def clips():
    return np.stack([R.synthetic_clip(0, 30.0), R.synthetic_clip(1, 5.0)])


@pytest.fixture(scope="module")
def micro(clips):
    W = R.synthetic_weights(MICRO, seed=7)
    mels = np.stack([R.log_mel_spectrogram(a) for a in clips])
    with torch.no_grad():
        xa = R.encoder_forward(W, MICRO, torch.from_numpy(mels))
    return W, mels, xa


@pytest.mark.parametrize("n_mels", [80, 128])
def test_logmel_matches_oracle_and_golden(clips, golden_dir, n_mels):
    import whisper_ipa_amd as wipa

    mel = wipa.log_mel_spectrogram(clips, n_mels=n_mels)
    assert tuple(mel.shape) == (2, 3000, n_mels) and mel.dtype == torch.float32
    got = mel.cpu().numpy()
    for i, name in enumerate(("full", "short")):
        ref = R.log_mel_spectrogram(clips[i], n_mels)
        assert np.abs(got[i] - ref).max() < 1e-3, np.abs(got[i] - ref).max()
        g = np.load(os.path.join(golden_dir, "mel.npz"))
        rows = g[f"{name}_{n_mels}_rows"]
        assert np.abs(got[i][rows] - g[f"{name}_{n_mels}_slices"]).max() < 1e-3
    single = wipa.log_mel_spectrogram(clips[0], n_mels=n_mels)
    assert tuple(single.shape) == (3000, n_mels)
    assert torch.equal(single, mel[0])


def test_logmel_padded_layout(clips):
    from whisper_ipa_amd import audio

    a = torch.from_numpy(clips).cuda()
    p = audio.log_mel_padded(a, 80, torch.bfloat16)
    assert tuple(p.shape) == (2 * 3002 + 4, 80)
    v = p[: 2 * 3002].view(2, 3002, 80).float()
    assert (v[:, 0] == 0).all() and (v[:, 3001] == 0).all() and (p[2 * 3002 :] == 0).all()
    ref = R.log_mel_spectrogram(clips[1])
    assert np.abs(v[1, 1:3001].cpu().numpy() - ref).max() < 2e-2  # bf16 storage


def test_encoder_f32_matches_oracle_and_golden(micro, golden_dir):
    W, mels, xa = micro
    m = _model(MICRO, W, torch.float32)
    feats = m.encoder(torch.from_numpy(mels).cuda())
    assert tuple(feats.shape) == (2, 1500, 128)
    err = (feats.cpu() - xa).abs().max().item()
    assert err < 1e-3, err
    g = np.load(os.path.join(golden_dir, "micro_model.npz"))
    assert np.abs(feats.cpu().numpy()[:, g["enc_rows"]] - g["enc_slices"]).max() < 1e-3


def test_logits_and_loss_f32_match_oracle_and_golden(micro, golden_dir):
    from whisper_ipa_amd import ops

    W, mels, xa = micro
    g = np.load(os.path.join(golden_dir, "micro_model.npz"))
    tokens = torch.from_numpy(g["tokens"])
    m = _model(MICRO, W, torch.float32)
    logits = m.logits(tokens[:, :-1].cuda(), xa.cuda())
    with torch.no_grad():
        ref = R.decoder_forward(W, MICRO, tokens[:, :-1], xa)
        ref_loss = R.loss_from_features(W, MICRO, xa, tokens, 50257)
    err = (logits.cpu() - ref).abs().max().item()
    assert err < 1e-3, err  # north_star: logits within 1e-3 in fp32
    assert np.abs(logits.cpu().numpy()[:, :, g["logit_cols"]] - g["logit_slices"]).max() < 2e-3
    B, T = tokens.shape[0], tokens.shape[1] - 1
    flat = logits.reshape(B * T, -1)
    base = logits.as_strided((B * T, logits.stride(1)), (logits.stride(1), 1))
    out, _ = ops.masked_ce(base, tokens.to(torch.int32).cuda(), MICRO.n_vocab, 50257)
    loss = float(out[0] / out[1].clamp(min=1))
    assert abs(loss - float(ref_loss)) < 1e-3, (loss, float(ref_loss))
    assert abs(loss - float(g["loss"][0])) < 1e-3


@pytest.mark.parametrize("use_graph", [False, True])
def test_greedy_f32_bit_exact_vs_oracle_and_golden(micro, golden_dir, use_graph):
    from whisper_ipa_amd.decoding import greedy_decode_tokens

    W, mels, xa = micro
    sp = R.SpecialTokens.multilingual()
    always, first = R.suppress_lists(sp)
    init = list(sp.sot_sequence_including_notimestamps(0))
    m = _model(MICRO, W, torch.float32)
    res = greedy_decode_tokens(m, xa.cuda(), init, always, first, sp.eot, max_new_tokens=24, stop_on_eot=False,
                               use_graph=use_graph)
    with torch.no_grad():
        ref = R.greedy_decode(W, MICRO, xa, init, always, first, sp.eot, sample_len=24, stop_on_eot=False)
    assert res.tokens.shape == ref.tokens.shape
    assert (res.tokens == ref.tokens).all(), (res.tokens.tolist(), ref.tokens.tolist())
    assert np.abs(res.sum_logprobs - ref.sum_logprobs).max() < 1e-2
    g = np.load(os.path.join(golden_dir, "micro_model.npz"))
    n = g["greedy_tokens"].shape[1]
    assert (res.tokens[:, :n] == g["greedy_tokens"]).all()


def test_greedy_eot_latch_and_early_stop():
    from whisper_ipa_amd.decoding import greedy_decode_tokens

    dims = R.ModelDimensions(80, 1500, 64, 1, 1, 51865, 448, 64, 1, 1)
    W = R.synthetic_weights(dims, seed=3)
    torch.manual_seed(0)
    xa = torch.randn(3, 1500, 64)
    sp = R.SpecialTokens.multilingual()
    allowed = {sp.eot, 100, 200, 300}
    always = [t for t in range(dims.n_vocab) if t not in allowed]
    init = list(sp.sot_sequence_including_notimestamps(0))
    m = _model(dims, W, torch.float32)
    res = greedy_decode_tokens(m, xa.cuda(), init, always, [], sp.eot, max_new_tokens=40)
    with torch.no_grad():
        ref = R.greedy_decode(W, dims, xa, init, always, [], sp.eot, sample_len=40)
    assert res.n_steps == ref.n_steps
    assert (res.tokens == ref.tokens).all(), (res.tokens.tolist(), ref.tokens.tolist())
    assert np.abs(res.sum_logprobs - ref.sum_logprobs).max() < 1e-3


def test_greedy_full_length_and_batch_of_one(micro):
    """Edge cases of the decode loop: the reference's maximum sample length (n_text_ctx // 2 = 224 new tokens, the
    self-KV cache crossing every wave-split / tail boundary of the decode attention) and a batch of ONE clip, both
    bit-exact against the oracle's greedy loop."""
    from whisper_ipa_amd.decoding import greedy_decode_tokens

    W, mels, xa = micro
    sp = R.SpecialTokens.multilingual()
    always, first = R.suppress_lists(sp)
    init = list(sp.sot_sequence_including_notimestamps(0))
    m = _model(MICRO, W, torch.float32)
    n = MICRO.n_text_ctx // 2
    res = greedy_decode_tokens(m, xa.cuda(), init, always, first, sp.eot, max_new_tokens=n, stop_on_eot=False)
    with torch.no_grad():
        ref = R.greedy_decode(W, MICRO, xa, init, always, first, sp.eot, sample_len=n, stop_on_eot=False)
    assert res.tokens.shape == (2, 4 + n) == ref.tokens.shape
    gate = np.cumprod(ref.margins > 1e-3, axis=1).astype(bool)  # compare up to the first fp32 near-tie, if any
    assert gate[:, :64].all()
    assert (res.tokens[:, 4:][gate] == ref.tokens[:, 4:][gate]).all()
    one = greedy_decode_tokens(m, xa[:1].cuda(), init, always, first, sp.eot, max_new_tokens=32, stop_on_eot=False)
    assert (one.tokens[0] == res.tokens[0, : 4 + 32]).all()


def test_teacher_forced_rows_longer_than_the_text_context_are_refused(micro):
    """ipa_data_loader.py:124-131 does not truncate, so a row can exceed n_text_ctx = 448; the reference then fails on
    the positional-embedding slice.  Here the C ABI reports it instead of reading past the table."""
    from whisper_ipa_amd._lib import WipaError

    W, mels, xa = micro
    m = _model(MICRO, W, torch.float32)
    ok = torch.randint(0, 50000, (1, MICRO.n_text_ctx), dtype=torch.int64).cuda()
    assert m.logits(ok, xa[:1].cuda()).shape == (1, MICRO.n_text_ctx, MICRO.n_vocab)
    too_long = torch.randint(0, 50000, (1, MICRO.n_text_ctx + 1), dtype=torch.int64).cuda()
    with pytest.raises((WipaError, ValueError)):
        m.logits(too_long, xa[:1].cuda())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_prompt_prefill_equals_stepwise_prompt(micro, monkeypatch, dtype):
    """wipa_decoder_prefill (the prompt in one batched pass, cross K/V streamed once for all prompt positions) against the
    position-by-position prompt: same token ids, same sum of log-probabilities, in f32 and in bf16."""
    from whisper_ipa_amd.decoding import greedy_decode_tokens

    W, mels, xa = micro
    sp = R.SpecialTokens.multilingual()
    always, first = R.suppress_lists(sp)
    init = list(sp.sot_sequence_including_notimestamps(0))
    m = _model(MICRO, W, dtype)
    feats = xa.cuda().to(dtype)
    a = greedy_decode_tokens(m, feats, init, always, first, sp.eot, max_new_tokens=24, stop_on_eot=False)
    monkeypatch.setenv("WIPA_NO_PREFILL", "1")
    b = greedy_decode_tokens(m, feats, init, always, first, sp.eot, max_new_tokens=24, stop_on_eot=False)
    assert a.tokens.shape == b.tokens.shape == (2, 4 + 24)
    assert (a.tokens == b.tokens).all(), (a.tokens.tolist(), b.tokens.tolist())
    assert np.abs(a.sum_logprobs - b.sum_logprobs).max() < (1e-3 if dtype == torch.float32 else 0.3)
    monkeypatch.delenv("WIPA_NO_PREFILL")
    for n_init in (2, 3):  # shorter prompts take the same path
        c = greedy_decode_tokens(m, feats, init[:n_init], always, first, sp.eot, max_new_tokens=6, stop_on_eot=False)
        monkeypatch.setenv("WIPA_NO_PREFILL", "1")
        d = greedy_decode_tokens(m, feats, init[:n_init], always, first, sp.eot, max_new_tokens=6, stop_on_eot=False)
        monkeypatch.delenv("WIPA_NO_PREFILL")
        assert (c.tokens == d.tokens).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_decode_step_equals_unfused_step(micro, small2, monkeypatch, dtype):
    """The fused decode step (csrc/decode_fused.hip: 5 launches per layer) against the unfused one (11 launches per layer,
    WIPA_DECODE_FUSED=0): same ids, same log-probabilities, same last-step logits up to summation order -- on the micro model
    and at whisper-small width, graph replay and eager."""
    from whisper_ipa_amd.decoding import greedy_decode_tokens

    sp = R.SpecialTokens.multilingual()
    always, first = R.suppress_lists(sp)
    init = list(sp.sot_sequence_including_notimestamps(0))
    for dims, (W, mels, xa), n_new in ((MICRO, micro, 40), (SMALL2, small2, 12)):
        m = _model(dims, W, dtype, cross_attention="cached")  # the step variants of the cached-K/V form
        feats = xa.cuda().to(dtype)
        out = {}
        for fused in ("1", "2", "0"):
            monkeypatch.setenv("WIPA_DECODE_FUSED", fused)
            for use_graph in (True, False):
                r = greedy_decode_tokens(m, feats, init, always, first, sp.eot, max_new_tokens=n_new, stop_on_eot=False,
                                         use_graph=use_graph)
                out[(fused, use_graph)] = (r.tokens, r.sum_logprobs.copy(), r.last_logits.float().cpu().clone())
        monkeypatch.delenv("WIPA_DECODE_FUSED")
        for mode in ("1", "2"):
            assert (out[(mode, True)][0] == out[(mode, False)][0]).all() and torch.equal(out[(mode, True)][2], out[(mode, False)][2])  # graph == eager
        h, b = out[("2", True)], out[("0", True)]  # mode 2: only the cross block differs from the unfused step
        if dtype == torch.float32:
            assert (h[0] == b[0]).all() and np.abs(h[1] - b[1]).max() < 1e-3 and (h[2] - b[2]).abs().max() < 2e-4
        else:
            same2 = np.cumprod(h[0] == b[0], axis=1).astype(bool)
            assert same2[:, : 4 + 6].all(), (h[0].tolist(), b[0].tolist())
        a = out[("1", True)]
        if dtype == torch.float32:
            assert (a[0] == b[0]).all(), (a[0].tolist(), b[0].tolist())
            assert np.abs(a[1] - b[1]).max() < 1e-3 and (a[2] - b[2]).abs().max() < 2e-4
        else:
            # bf16 rounding points are the same but the f32 summation order inside a projection differs, which can move a
            # stored bf16 value by one ulp: ids agree up to the first near-tie
            same = np.cumprod(a[0] == b[0], axis=1).astype(bool)
            assert same[:, : 4 + 6].all(), (a[0].tolist(), b[0].tolist())
            rows = same.all(axis=1)
            if rows.any():
                assert (a[2][rows] - b[2][rows]).abs().max() < 0.25


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_step_tail_is_bit_identical_to_the_separate_launches(micro, small2, monkeypatch, dtype):
    """Round 4: a decode step ends with ONE launch -- greedy update + embedding of the chosen token + first LayerNorm of the next
    position + position advance (wipa_greedy_step_embed; wipa_embed_layernorm once before the first step of a run) -- instead
    of greedy_step, advance_pos, embed and add_slabs_layernorm (WIPA_DECODE_TAIL=0).  Same arithmetic in the same order: token
    ids, log-probability sums and last logits must be BIT-identical, with the batched prompt pass and with the prompt walked
    position by position, graph replay and eager, several run() calls per decode (check_every), a forced history, a bare
    [sot] prompt, both cross-attention forms and fp8 tables."""
    from whisper_ipa_amd.decoding import forced_decode_logits, greedy_decode_tokens

    sp = R.SpecialTokens.multilingual()
    always, first = R.suppress_lists(sp)
    init = list(sp.sot_sequence_including_notimestamps(0))
    cases = [(MICRO, micro, "auto", False)]
    if dtype == torch.bfloat16:
        cases += [(SMALL2, small2, "absorbed", False), (SMALL2, small2, "cached", False), (SMALL2, small2, "auto", True)]
    # the row-scanning tail: the logits projection that carries the greedy partials (another summation order of the log-prob;
    # test_logits_projection_with_greedy_partials_*) is switched off for both sides
    monkeypatch.setenv("WIPA_LOGITS_FUSED", "0")
    for dims, (W, mels, xa), cross, fp8 in cases:
        m = _model(dims, W, dtype, cross_attention=cross)
        if fp8:
            m.quantize_weights()
        feats = xa.cuda().to(dtype)
        out = {}
        for tail in ("1", "0"):
            monkeypatch.setenv("WIPA_DECODE_TAIL", tail)
            res = []
            for use_graph in (True, False):
                r = greedy_decode_tokens(m, feats, init, always, first, sp.eot, max_new_tokens=21, stop_on_eot=False, use_graph=use_graph)
                res.append((r.tokens, r.sum_logprobs.copy(), r.last_logits.float().cpu().clone()))
            r = greedy_decode_tokens(m, feats, init, always, first, sp.eot, max_new_tokens=21, stop_on_eot=True, check_every=3)
            res.append((r.tokens, r.sum_logprobs.copy(), None))
            monkeypatch.setenv("WIPA_NO_PREFILL", "1")
            r = greedy_decode_tokens(m, feats, init, always, first, sp.eot, max_new_tokens=9, stop_on_eot=False)
            res.append((r.tokens, r.sum_logprobs.copy(), r.last_logits.float().cpu().clone()))
            monkeypatch.delenv("WIPA_NO_PREFILL")
            r = greedy_decode_tokens(m, feats, init[:1], always, first, sp.eot, max_new_tokens=5, stop_on_eot=False)  # bare [sot]
            res.append((r.tokens, r.sum_logprobs.copy(), r.last_logits.float().cpu().clone()))
            hist = res[0][0][:, : 4 + 10].copy()
            hist[:, 6] = 1000 + np.arange(hist.shape[0])  # a history the path would not have chosen itself
            tr, chosen = forced_decode_logits(m, feats, hist, 4, always, first, sp.eot)
            res.append((chosen, np.zeros(1), tr.float().cpu().clone()))
            out[tail] = res
        monkeypatch.delenv("WIPA_DECODE_TAIL")
        for k, (a, b) in enumerate(zip(out["1"], out["0"])):
            assert a[0].shape == b[0].shape and (a[0] == b[0]).all(), (dims.n_text_state, cross, fp8, k, a[0].tolist(), b[0].tolist())
            assert np.array_equal(a[1], b[1]), (cross, fp8, k)
            if a[2] is not None:
                assert torch.equal(a[2], b[2]), (cross, fp8, k)
        g, e = out["1"][0], out["1"][1]
        assert (g[0] == e[0]).all() and torch.equal(g[2], e[2])  # graph == eager


def test_logits_projection_with_greedy_partials_keeps_ids_and_last_logits(small2, monkeypatch):
    """Round 4: the decode step's logits projection keeps per-(row, wave) max / arg-max / sum-exp partials of the filtered logits
    (wipa_logits_greedy) and the step's last launch merges them (wipa_greedy_step_embed_partials) instead of reading the logits
    back; every step of a run() call but the last does not write its logits at all.  Against WIPA_LOGITS_FUSED=0 (plain GEMM +
    row-scanning tail): token ids identical, the logits of the last step and the log-probability sums equal up to the summation
    order (1e-5 relative; bit-identical logits at > 32 rows: tests/test_gpu_kernels.py) -- graph and eager, one run() call and several (check_every), the prompt walked step by step,
    a bare [sot] prompt, a forced history (every step's logits are read: all must be written), both cross-attention forms; and a
    clip's ids / log-probs do not depend on the batch it rides in."""
    from whisper_ipa_amd.decoding import forced_decode_logits, greedy_decode_tokens

    W, mels, xa = small2
    sp = R.SpecialTokens.multilingual()
    always, first = R.suppress_lists(sp)
    init = list(sp.sot_sequence_including_notimestamps(0))
    feats = xa.cuda().to(torch.bfloat16)
    for cross in ("absorbed", "cached"):
        m = _model(SMALL2, W, torch.bfloat16, cross_attention=cross)
        out = {}
        for fused in ("1", "0"):
            monkeypatch.setenv("WIPA_LOGITS_FUSED", fused)
            res = []
            for use_graph in (True, False):
                r = greedy_decode_tokens(m, feats, init, always, first, sp.eot, max_new_tokens=21, stop_on_eot=False, use_graph=use_graph)
                res.append((r.tokens, r.sum_logprobs.copy(), r.last_logits.float().cpu().clone()))
            r = greedy_decode_tokens(m, feats, init, always, first, sp.eot, max_new_tokens=21, stop_on_eot=True, check_every=3)
            res.append((r.tokens, r.sum_logprobs.copy(), None))
            monkeypatch.setenv("WIPA_NO_PREFILL", "1")
            r = greedy_decode_tokens(m, feats, init, always, first, sp.eot, max_new_tokens=9, stop_on_eot=False)
            res.append((r.tokens, r.sum_logprobs.copy(), r.last_logits.float().cpu().clone()))
            monkeypatch.delenv("WIPA_NO_PREFILL")
            r = greedy_decode_tokens(m, feats, init[:1], always, first, sp.eot, max_new_tokens=5, stop_on_eot=False)  # bare [sot]
            res.append((r.tokens, r.sum_logprobs.copy(), r.last_logits.float().cpu().clone()))
            hist = res[0][0][:, : 4 + 10].copy()
            hist[:, 6] = 1000 + np.arange(hist.shape[0])
            tr, chosen = forced_decode_logits(m, feats, hist, 4, always, first, sp.eot)
            res.append((chosen, np.zeros(1), tr.float().cpu().clone()))
            one = greedy_decode_tokens(m, feats[1:2].contiguous(), init, always, first, sp.eot, max_new_tokens=21, stop_on_eot=False)
            res.append((one.tokens, one.sum_logprobs.copy(), None))
            out[fused] = res
        monkeypatch.delenv("WIPA_LOGITS_FUSED")
        for k, (a, b) in enumerate(zip(out["1"], out["0"])):
            assert a[0].shape == b[0].shape and (a[0] == b[0]).all(), (cross, k, a[0].tolist(), b[0].tolist())
            assert np.allclose(a[1], b[1], rtol=1e-5, atol=1e-5), (cross, k, a[1], b[1])
            if a[2] is not None:  # (a few rows: the plain path's GEMM splits K over waves -- equal up to the summation order)
                fin = torch.isfinite(b[2])
                assert torch.equal(fin, torch.isfinite(a[2])) and torch.allclose(a[2][fin], b[2][fin], rtol=1e-5, atol=1e-5), (cross, k)
        # graph == eager bit for bit, and clip 1 alone == clip 1 in the batch (ids and log-prob sum)
        g, e, alone = out["1"][0], out["1"][1], out["1"][-1]
        assert (g[0] == e[0]).all() and np.array_equal(g[1], e[1]) and torch.equal(g[2], e[2])
        assert (alone[0][0] == g[0][1]).all() and alone[1][0] == g[1][1]


def test_absorbed_cross_block_fused_prologue_equals_separate_launches(small2, monkeypatch):
    """The absorbed cross block of the decode step: [slab sum + LayerNorm + cross query + absorbed query] in ONE launch
    (cross_absorb_prologue_kernel, the default) against the three separate launches (WIPA_ABS_FUSED_PROLOGUE=0:
    add_slabs_layernorm, query GEMM, cross_absorb_q).  Same bf16 rounding points; the f32 summation orders differ (LayerNorm
    statistics per wave instead of per block, K split over eight waves), which can move a stored bf16 value by one ulp: the
    logits along a fixed token history agree to a small fraction of their spread, graph replay == eager bit for bit."""
    from whisper_ipa_amd.decoding import forced_decode_logits, greedy_decode_tokens

    W, mels, xa = small2
    sp = R.SpecialTokens.multilingual()
    always, first = R.suppress_lists(sp)
    init = list(sp.sot_sequence_including_notimestamps(0))
    m = _model(SMALL2, W, torch.bfloat16, cross_attention="absorbed")
    feats = xa.cuda().to(torch.bfloat16)
    base = greedy_decode_tokens(m, feats, init, always, first, sp.eot, max_new_tokens=16, stop_on_eot=False)
    traces = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("WIPA_ABS_FUSED_PROLOGUE", flag)
        t_graph, _ = forced_decode_logits(m, feats, base.tokens, 4, always, first, sp.eot, use_graph=True)
        t_eager, _ = forced_decode_logits(m, feats, base.tokens, 4, always, first, sp.eot, use_graph=False)
        assert torch.equal(t_graph, t_eager), flag
        traces[flag] = t_graph.float().cpu()
    monkeypatch.delenv("WIPA_ABS_FUSED_PROLOGUE")
    finite = torch.isfinite(traces["1"]) & torch.isfinite(traces["0"])
    diff = (traces["1"] - traces["0"])[finite].abs().max().item()
    spread = traces["0"][finite].std().item()
    print(f"\nabsorbed cross block, fused prologue vs separate launches: max logit difference {diff:.4f} = {diff / spread:.4f} of the logit std")
    assert diff < 0.03 * spread, (diff, spread)


def test_cross_splits_is_a_model_setting_that_keeps_graph_eager_and_batch_invariance(small2):
    """Round 4: Whisper.cross_splits -> wipa_model_cfg.dec_cross_splits: the frame splits of the decode step's streaming launch
    (2 = half-chip launches for several passes in flight).  For every count: the step graph is re-captured under its own key and
    replays what the eager step computes, bit for bit; a clip's ids and logits do not depend on the batch it rides in; the
    logits along a fixed history stay within a small fraction of their spread of the default's (only the order of the softmax
    merges moves); 0 and 4 are the same setting; the setting survives a weight reload; and the runtime refuses counts outside
    0..4 or a count on the cached form."""
    import ctypes as C

    from whisper_ipa_amd import _lib
    from whisper_ipa_amd.decoding import forced_decode_logits, greedy_decode_tokens

    W, mels, xa = small2
    sp = R.SpecialTokens.multilingual()
    always, first = R.suppress_lists(sp)
    init = list(sp.sot_sequence_including_notimestamps(0))
    m = _model(SMALL2, W, torch.bfloat16, cross_attention="absorbed")
    feats = xa.cuda().to(torch.bfloat16)
    base = greedy_decode_tokens(m, feats, init, always, first, sp.eot, max_new_tokens=12, stop_on_eot=False)
    t4, _ = forced_decode_logits(m, feats, base.tokens, 4, always, first, sp.eot, use_graph=True)
    traces = {}
    for n in (0, 1, 2, 3, 4):
        m.cross_splits = n
        assert m.packed(absorbed=True)["cfg"].dec_cross_splits == n
        tg, _ = forced_decode_logits(m, feats, base.tokens, 4, always, first, sp.eot, use_graph=True)
        te, _ = forced_decode_logits(m, feats, base.tokens, 4, always, first, sp.eot, use_graph=False)
        assert torch.equal(tg, te), n
        one, _ = forced_decode_logits(m, feats[1:2].contiguous(), base.tokens[1:2], 4, always, first, sp.eot, use_graph=True)
        assert torch.equal(one[0], tg[1]), n  # clip 1 alone == clip 1 in the batch
        traces[n] = tg.float().cpu()
    assert torch.equal(traces[0], traces[4]) and torch.equal(traces[4], t4.float().cpu())
    fin = torch.isfinite(traces[4])
    spread = traces[4][fin].std().item()
    for n in (1, 2, 3):
        diff = (traces[n] - traces[4])[fin].abs().max().item()
        print(f"\ncross_splits {n} vs 4: max logit difference {diff:.4f} = {diff / spread:.4f} of the logit std")
        assert diff < 0.03 * spread, (n, diff, spread)
    m.cross_splits = 2
    m.load_weights(W)  # repacks: the setting is the model's, not the packed table's
    assert m.packed(absorbed=True)["cfg"].dec_cross_splits == 2
    with pytest.raises(_lib.WipaError):
        m.cross_splits = 5
    pk = m.packed(absorbed=True)
    bad = type(pk["cfg"]).from_buffer_copy(pk["cfg"])
    bad.dec_cross_splits = 7
    assert _lib.lib().wipa_decoder_layout(C.byref(bad), 4, C.byref(_lib.DecLayout())) != 0
    cached = type(pk["cfg"]).from_buffer_copy(m.packed(absorbed=False)["cfg"])
    cached.dec_cross_splits = 2
    assert _lib.lib().wipa_decoder_layout(C.byref(cached), 4, C.byref(_lib.DecLayout())) != 0


@pytest.mark.parametrize("cross_splits", [0, 2])
def test_prompt_prefill_equals_stepwise_prompt_absorbed(small2, monkeypatch, cross_splits):
    """wipa_decoder_prefill and wipa_decoder_run take the SAME frame-split count from wipa_model_cfg.dec_cross_splits
    (include/wipa.h), so on the absorbed form (whisper-small width, bf16) the batched prompt pass followed by steps gives the
    ids of the position-by-position prompt in the library default AND in the several-passes-in-flight setting (2 splits);
    the step logits after the prompt agree to bf16 rounding (the prompt pass batches 4 positions into one GEMM)."""
    from whisper_ipa_amd.decoding import greedy_decode_tokens

    W, mels, xa = small2
    sp = R.SpecialTokens.multilingual()
    always, first = R.suppress_lists(sp)
    init = list(sp.sot_sequence_including_notimestamps(0))
    m = _model(SMALL2, W, torch.bfloat16, cross_attention="absorbed")
    m.cross_splits = cross_splits
    feats = xa.cuda().to(torch.bfloat16)
    a = greedy_decode_tokens(m, feats, init, always, first, sp.eot, max_new_tokens=16, stop_on_eot=False)
    la = a.last_logits.float().cpu().clone()
    monkeypatch.setenv("WIPA_NO_PREFILL", "1")
    b = greedy_decode_tokens(m, feats, init, always, first, sp.eot, max_new_tokens=16, stop_on_eot=False)
    lb = b.last_logits.float().cpu().clone()
    monkeypatch.delenv("WIPA_NO_PREFILL")
    assert (a.tokens == b.tokens).all(), (cross_splits, a.tokens.tolist(), b.tokens.tolist())
    fin = torch.isfinite(la) & torch.isfinite(lb)
    diff, spread = (la - lb)[fin].abs().max().item(), lb[fin].std().item()
    print(f"\nprefill + steps vs stepwise prompt, absorbed, cross_splits {cross_splits}: last-step logits differ by {diff:.4f} = {diff / spread:.4f} of their std")
    assert diff < 0.03 * spread, (diff, spread)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", ["0", "1", "2"])
def test_decode_never_reads_unwritten_state(micro, monkeypatch, dtype, mode):
    """The decode blob comes from torch.empty: whatever the caches and scratch held before must not reach the result.  The
    blob is filled with 0xFF bytes (NaN in f32 and bf16, -1 as a token id) before the run, with and without the batched
    prompt pass, for every step variant; the ids and log-probabilities must equal those of a run on a zeroed blob."""
    from whisper_ipa_amd.decoding import _state_for, greedy_decode_tokens
    from whisper_ipa_amd.runtime import on_stream

    W, mels, xa = micro
    sp = R.SpecialTokens.multilingual()
    always, first = R.suppress_lists(sp)
    init = list(sp.sot_sequence_including_notimestamps(0))
    monkeypatch.setenv("WIPA_DECODE_FUSED", mode)
    m = _model(MICRO, W, dtype)
    feats = xa.cuda().to(dtype)
    out = {}
    for prompt in (init, init[:1]):  # 4-token prompt (prefill pass) and a bare [sot] (the first step runs at position 0)
        for fill in (0x00, 0xFF):
            with on_stream():
                _state_for(m, feats.shape[0]).blob.fill_(fill)
            r = greedy_decode_tokens(m, feats, prompt, always, first, sp.eot, max_new_tokens=10, stop_on_eot=False)
            out[(len(prompt), fill)] = (r.tokens, r.sum_logprobs.copy())
        a, b = out[(len(prompt), 0x00)], out[(len(prompt), 0xFF)]
        assert np.isfinite(b[1]).all() and (a[0] == b[0]).all() and np.array_equal(a[1], b[1]), (mode, len(prompt), a[0].tolist(), b[0].tolist())


def test_decode_blob_follows_the_layout_after_quantize_and_cross_attention_changes(small2):
    """the blob layout depends on the configuration.  A bf16 model that has decoded with the absorbed
    cross-attention owns a blob with ONE copy of the features; quantize_weights() (fp8 tables -> cached K / V) or a switch to
    cross_attention="cached" needs K and V of every layer.  The state must be re-made (decoding._state_for compares the
    layouts), the result must equal a FRESH model in the new configuration, and the C ABI must refuse a blob that is too small."""
    import ctypes as C

    from whisper_ipa_amd import _lib
    from whisper_ipa_amd.decoding import _state_for, greedy_decode_tokens
    from whisper_ipa_amd.runtime import on_stream, ptr, sptr

    W, mels, xa = small2
    sp = R.SpecialTokens.multilingual()
    always, first = R.suppress_lists(sp)
    init = list(sp.sot_sequence_including_notimestamps(0))
    feats = xa.cuda().to(torch.bfloat16)

    def run(m):
        return greedy_decode_tokens(m, feats, init, always, first, sp.eot, max_new_tokens=8, stop_on_eot=False)

    m = _model(SMALL2, W, torch.bfloat16)
    assert m.cross_absorbed
    run(m)
    small_blob = _state_for(m, 2).blob
    small_bytes = small_blob.numel()
    # (a) the same model, now with fp8 tables: cached K / V, a much larger blob
    m.quantize_weights()
    assert not m.cross_absorbed
    got = run(m)
    assert _state_for(m, 2).blob.numel() > 1.5 * small_bytes  # K and V of both layers against one copy of the features
    fresh = _model(SMALL2, W, torch.bfloat16)
    fresh.quantize_weights()
    want = run(fresh)
    assert (got.tokens == want.tokens).all() and np.array_equal(got.sum_logprobs, want.sum_logprobs)
    # (b) absorbed -> cached by attribute on a model that has already decoded
    m2 = _model(SMALL2, W, torch.bfloat16)
    a = run(m2)
    m2.cross_attention = "cached"
    m2._invalidate()
    b = run(m2)
    c = run(_model(SMALL2, W, torch.bfloat16, cross_attention="cached"))
    assert (b.tokens == c.tokens).all() and np.array_equal(b.sum_logprobs, c.sum_logprobs)
    assert a.tokens.shape == b.tokens.shape
    # (d) cross_attention="auto" decides per call from (batch, new tokens) by the measured table: long outputs at a small batch
    # take the cached form (profiles/r03_cached_vs_absorbed.txt: -4.6 % for absorbed at 64 clips x 224 tokens), everything else
    # the absorbed one; both forms live side by side on one model and give the ids of a model pinned to that form
    m3 = _model(SMALL2, W, torch.bfloat16)
    assert m3.use_absorbed(64, 64) and m3.use_absorbed(128, 224) and m3.use_absorbed() and not m3.use_absorbed(64, 224)
    assert not m3.use_absorbed(2, 192) and m3.use_absorbed(2, 191)
    short = greedy_decode_tokens(m3, feats, init, always, first, sp.eot, max_new_tokens=8, stop_on_eot=False)
    long_ = greedy_decode_tokens(m3, feats, init, always, first, sp.eot, max_new_tokens=200, stop_on_eot=False)
    short2 = greedy_decode_tokens(m3, feats, init, always, first, sp.eot, max_new_tokens=8, stop_on_eot=False)
    assert sorted(k[1] for k in m3._dec_states) == [0, 1]  # one blob per form
    ref_long = greedy_decode_tokens(_model(SMALL2, W, torch.bfloat16, cross_attention="cached"), feats, init, always, first, sp.eot,
                                    max_new_tokens=200, stop_on_eot=False)
    assert (long_.tokens == ref_long.tokens).all() and (short.tokens == a.tokens).all() and (short2.tokens == short.tokens).all()
    # (c) the ABI itself refuses the old blob under the new configuration (no launch, an error string)
    L = _lib.lib()
    pk = m2.packed()
    with on_stream() as s:
        rc = L.wipa_decoder_set_audio(C.byref(pk["cfg"]), pk["dec_tab"], ptr(feats), ptr(small_blob), small_bytes, 2, sptr(s))
    assert rc != 0 and b"state blob" in L.wipa_last_error()


def test_no_graph_replay_while_hardware_counters_are_attached(micro, monkeypatch):
    """rocprofv3 --pmc sets ROCPROF_COUNTER_COLLECTION; the runtime then enqueues the decode steps eagerly instead of capturing
    / replaying hipGraphs (runtime.hip counters_attached).  Same ids either way."""
    from whisper_ipa_amd.decoding import greedy_decode_tokens

    W, mels, xa = micro
    sp = R.SpecialTokens.multilingual()
    always, first = R.suppress_lists(sp)
    init = list(sp.sot_sequence_including_notimestamps(0))
    m = _model(MICRO, W, torch.float32)
    a = greedy_decode_tokens(m, xa.cuda(), init, always, first, sp.eot, max_new_tokens=12, stop_on_eot=False)
    monkeypatch.setenv("ROCPROF_COUNTER_COLLECTION", "1")
    m2 = _model(MICRO, W, torch.float32)  # fresh state: nothing captured before
    b = greedy_decode_tokens(m2, xa.cuda(), init, always, first, sp.eot, max_new_tokens=12, stop_on_eot=False)
    assert (a.tokens == b.tokens).all() and np.array_equal(a.sum_logprobs, b.sum_logprobs)


def test_detect_language_matches_oracle(micro):
    from whisper_ipa_amd.decoding import detect_language
    from whisper_ipa_amd.tokenizer import get_tokenizer

    W, mels, xa = micro
    m = _model(MICRO, W, torch.float32)
    tok = get_tokenizer(True)
    lang, probs = detect_language(m, xa.cuda(), tok)
    with torch.no_grad():
        ref = R.detect_language(W, MICRO, xa, R.SpecialTokens.multilingual())
    assert (np.asarray(lang) == ref).all()
    assert probs.shape == (2, 99) and abs(probs.sum(axis=1) - 1).max() < 1e-4


def test_decode_api_surface(micro):
    """transcribe_single.py:49-56 shape of use: features in, list of results with .text/.tokens."""
    import whisper_ipa_amd as wipa

    W, mels, xa = micro
    m = _model(MICRO, W, torch.float32)
    opts = wipa.DecodingOptions(language="en", without_timestamps=True, fp16=False, sample_len=6)
    out = wipa.decode(m, xa.cuda(), opts)
    assert isinstance(out, list) and len(out) == 2 and isinstance(out[0].text, str)
    one = m.decode(torch.from_numpy(mels[0]).cuda(), opts)
    assert isinstance(one, wipa.DecodingResult) and one.tokens == out[0].tokens
    sp = R.SpecialTokens.multilingual()
    always, first = R.suppress_lists(sp)
    with torch.no_grad():
        ref = R.greedy_decode(W, MICRO, xa, sp.sot_sequence_including_notimestamps(0), always, first, sp.eot, sample_len=6)
    for i in range(2):
        row = ref.tokens[i, 4:].tolist()
        if sp.eot in row:
            row = row[: row.index(sp.eot)]
        assert out[i].tokens == row


@pytest.fixture(scope="module")
def small2(clips):
    W = R.synthetic_weights(SMALL2, seed=11)
    mels = np.stack([R.log_mel_spectrogram(a) for a in clips])
    with torch.no_grad():
        xa = R.encoder_forward(W, SMALL2, torch.from_numpy(mels))
    return W, mels, xa


def test_small_width_f32_encoder_and_greedy(small2):
    """whisper-small width (d=768, 12 heads) exercises the full-size tiles in f32."""
    from whisper_ipa_amd.decoding import greedy_decode_tokens

    W, mels, xa = small2
    m = _model(SMALL2, W, torch.float32)
    feats = m.encoder(torch.from_numpy(mels).cuda())
    err = (feats.cpu() - xa).abs().max().item()
    assert err < 1e-3, err
    sp = R.SpecialTokens.multilingual()
    always, first = R.suppress_lists(sp)
    init = list(sp.sot_sequence_including_notimestamps(0))
    res = greedy_decode_tokens(m, feats, init, always, first, sp.eot, max_new_tokens=16, stop_on_eot=False)
    with torch.no_grad():
        ref = R.greedy_decode(W, SMALL2, xa, init, always, first, sp.eot, sample_len=16, stop_on_eot=False)
    assert (res.tokens == ref.tokens).all(), (res.tokens.tolist(), ref.tokens.tolist(), ref.margins.min())


@pytest.mark.parametrize("name,dims", [
    ("medium-width", R.ModelDimensions(80, 1500, 1024, 16, 1, 51865, 448, 1024, 16, 1)),
    ("large-v3-width", R.ModelDimensions(128, 1500, 1280, 20, 1, 51866, 448, 1280, 20, 1)),
])
def test_other_model_widths_f32(name, dims):
    """BASELINE configs 4/5 use whisper-medium (d=1024, 16 heads) and large-v3 (d=1280, 20 heads, 128 mels,
    51 866 tokens, 100 languages): one layer of each width through the same kernels, f32, against the oracle."""
    import whisper_ipa_amd as wipa
    from whisper_ipa_amd.decoding import greedy_decode_tokens

    W = R.synthetic_weights(dims, seed=21)
    audio = R.synthetic_clip(3, 30.0)[None]
    mel_ref = R.log_mel_spectrogram(audio[0], dims.n_mels)[None]
    with torch.no_grad():
        xa = R.encoder_forward(W, dims, torch.from_numpy(mel_ref))
    m = _model(dims, W, torch.float32)
    mel = wipa.log_mel_spectrogram(audio, n_mels=dims.n_mels)
    assert np.abs(mel.cpu().numpy() - mel_ref).max() < 1e-3
    feats = m.encoder(mel)
    err = (feats.cpu() - xa).abs().max().item()
    assert err < 1e-3, (name, err)
    sp = R.SpecialTokens.multilingual(100 if dims.n_vocab == 51866 else 99)
    assert m.num_languages == sp.n_langs
    always, first = R.suppress_lists(sp)
    init = list(sp.sot_sequence_including_notimestamps(0))
    res = greedy_decode_tokens(m, feats, init, always, first, sp.eot, max_new_tokens=10, stop_on_eot=False)
    with torch.no_grad():
        ref = R.greedy_decode(W, dims, xa, init, always, first, sp.eot, sample_len=10, stop_on_eot=False)
    assert (res.tokens == ref.tokens).all(), (name, res.tokens.tolist(), ref.tokens.tolist(), ref.margins.min())


@pytest.mark.parametrize("cross_attention", ["cached", "absorbed"])
def test_full_size_bench_workload_properties(cross_attention):
    """BASELINE configs[1] at FULL size (whisper-small 12+12 layers, bf16, 64 clips x 30 s): properties that do
    not need the (too slow) CPU oracle at this size --
      * batch invariance: a clip's features and token ids do not depend on which batch it rides in
        (clips 0..3 alone == rows 0..3 of the 64-clip batch, bit for bit);
      * determinism: two runs give identical ids;
      * prompt echo and vocabulary range; suppressed ids never appear."""
    import bench
    from whisper_ipa_amd import audio as A
    from whisper_ipa_amd.decoding import greedy_decode_tokens
    from whisper_ipa_amd.whisper import Whisper

    dims, W = bench.synthetic_weights_small(0)
    m = Whisper(dims, dtype=torch.bfloat16, cross_attention=cross_attention)
    m.load_weights(W)
    del W
    init, always, first, eot = bench.decode_setup()
    audio = torch.from_numpy(bench.synthetic_audio(0, 64)).cuda()
    audio[5, 16000 * 5:] = 0  # one short clip + zero padding, like the dataset's <= 6 s clips

    def run(a):
        mel = A.log_mel_padded(a, dims.n_mels, torch.bfloat16)
        feats = m.encode_padded(mel, a.shape[0])
        res = greedy_decode_tokens(m, feats, init, always, first, eot, max_new_tokens=16, stop_on_eot=False)
        return feats.float().cpu(), res.tokens

    f64, t64 = run(audio)
    f64b, t64b = run(audio)
    assert torch.equal(f64, f64b) and (t64 == t64b).all()
    f4, t4 = run(audio[:4].contiguous())
    assert torch.equal(f4, f64[:4]), (f4 - f64[:4]).abs().max()
    assert (t4 == t64[:4]).all()
    assert t64.shape == (64, 4 + 16) and (t64[:, :4] == np.array(init)).all()
    body = t64[:, 4:]
    assert body.min() >= 0 and body.max() < dims.n_vocab
    assert not np.isin(body, np.array(always)).any()
    assert not np.isin(body[:, 0], np.array(first)).any()
    assert torch.isfinite(f64).all() and len({tuple(r) for r in body.tolist()}) > 8  # rows differ (audio dependence)
    # a batch beyond the 64-row skinny-GEMM group (decode projections ride on grid.y): rows 0..63 unchanged
    audio96 = torch.cat([audio, torch.from_numpy(bench.synthetic_audio(64, 32)).cuda()])
    f96, t96 = run(audio96)
    assert torch.equal(f96[:64], f64) and (t96[:64] == t64).all()
    assert len({tuple(r) for r in t96[64:, 4:].tolist()}) > 4
    # two batches decoded as one group of 128 rows give each clip the tokens it gets in its own batch
    feats = m.encode_padded(A.log_mel_padded(audio, dims.n_mels, torch.bfloat16), 64)
    grouped = greedy_decode_tokens(m, torch.cat([feats, feats]), init, always, first, eot, max_new_tokens=16, stop_on_eot=False).tokens
    assert grouped.shape[0] == 128 and (grouped[:64] == grouped[64:]).all()
    assert (grouped[:64, : 4 + 16] == t64).all()


@pytest.mark.parametrize("cross_attention", ["cached", "absorbed"])
def test_small_width_bf16_close_and_tokens_match_where_margin_allows(small2, cross_attention):
    """bf16 path (the bench configuration's arithmetic) against the f32 oracle: features within
    bf16 tolerance; greedy tokens must equal the oracle's except at a step whose oracle
    top-1 margin is within twice the bf16 logit error MEASURED at that (row, step) (tests/parity_util.py)."""
    W, mels, xa = small2
    m = _model(SMALL2, W, torch.bfloat16, cross_attention=cross_attention)
    feats = m.encoder(torch.from_numpy(mels).cuda())
    assert feats.dtype == torch.bfloat16
    rel = ((feats.float().cpu() - xa).abs().max() / xa.abs().max()).item()
    assert rel < 5e-2, rel
    sp = R.SpecialTokens.multilingual()
    always, first = R.suppress_lists(sp)
    init = list(sp.sot_sequence_including_notimestamps(0))
    with torch.no_grad():
        ref = R.greedy_decode(W, SMALL2, xa, init, always, first, sp.eot, sample_len=16, stop_on_eot=False, keep_logits=True)
    err, rep = check_low_precision_decode(m, feats, ref, init, always, first, sp.eot, f"small width bf16 ({cross_attention})")
    print(f"\nsmall width bf16 ({cross_attention} cross-attention): max logit error {rep['max_logit_err']:.4f} = {rep['rel_err']:.4f} of the logit std, token match {rep['token_match']:.3f}")


@pytest.mark.parametrize("name,dims,n_new", [
    ("micro", MICRO, 24),
    ("small-width", SMALL2, 12),
    ("large-v3-width", R.ModelDimensions(128, 1500, 1280, 20, 1, 51866, 448, 1280, 20, 2), 10),
])
def test_fp8_weight_model_matches_dequantised_reference(name, dims, n_new):
    """BASELINE.json configs[4] (fp8-weight inference): a bf16 model whose matrices were quantised to e4m3 (per-row
    power-of-two scales) -- decode-step matrices streamed as 1-byte codes by the fp8 weight-streaming GEMM, everything else
    as the exact bf16 dequantisation -- against (a) the SAME model run entirely on the dequantised bf16 weights (features
    bit-identical, last-step logits equal up to f32 summation order, ids equal) and (b) the f32 oracle on the dequantised
    weights (measured logit error, tests/parity_util.py)."""
    import whisper_ipa_amd as wipa
    from whisper_ipa_amd.decoding import greedy_decode_tokens

    W = R.synthetic_weights(dims, seed=31)
    audio = np.stack([R.synthetic_clip(3, 30.0), R.synthetic_clip(4, 6.0)])
    m8 = _model(dims, W, torch.bfloat16)
    m8.quantize_weights("fp8_e4m3")
    assert m8.weights_format == "fp8_e4m3" and m8.packed()["cfg"].dec_w_dtype == 2
    Wdq = {k: v.float().cpu() for k, v in m8.flat_parameters().items()}
    changed = [k for k in W if W[k].dim() >= 2 and "positional" not in k]
    assert all(not torch.equal(Wdq[k], W[k]) for k in changed[:5])  # really quantised
    rel_q = max(float((Wdq[k] - W[k]).abs().max() / W[k].abs().max()) for k in changed)
    assert rel_q < 2.0 ** -4  # e4m3: 3 mantissa bits
    mref = _model(dims, Wdq, torch.bfloat16)
    mel = wipa.log_mel_spectrogram(audio, n_mels=dims.n_mels)
    f8, fr = m8.encoder(mel), mref.encoder(mel)
    assert torch.equal(f8, fr)  # the encoder multiplies by the same bf16 values
    sp = R.SpecialTokens.multilingual(100 if dims.n_vocab == 51866 else 99)
    always, first = R.suppress_lists(sp)
    init = list(sp.sot_sequence_including_notimestamps(0))
    a = greedy_decode_tokens(m8, f8, init, always, first, sp.eot, max_new_tokens=n_new, stop_on_eot=False)
    b = greedy_decode_tokens(mref, fr, init, always, first, sp.eot, max_new_tokens=n_new, stop_on_eot=False)
    same = np.cumprod(a.tokens == b.tokens, axis=1).astype(bool)
    assert same[:, : 4 + 4].all(), (name, a.tokens.tolist(), b.tokens.tolist())
    if same.all():
        dl = (a.last_logits.float() - b.last_logits.float()).abs().max().item()
        assert dl < 0.15, (name, dl)  # same rounding points; only the f32 summation order inside a projection differs
    # teacher-forced logits run on the dequantised table and agree with the reference model bit for bit
    toks = torch.from_numpy(a.tokens[:, :8]).cuda()
    assert torch.equal(m8.logits(toks, f8), mref.logits(toks, fr))
    # (b) the f32 oracle on the dequantised weights
    with torch.no_grad():
        mel_ref = np.stack([R.log_mel_spectrogram(x, dims.n_mels) for x in audio])
        xa = R.encoder_forward(Wdq, dims, torch.from_numpy(mel_ref))
        ref = R.greedy_decode(Wdq, dims, xa, init, always, first, sp.eot, sample_len=n_new, stop_on_eot=False, keep_logits=True)
    assert ((f8.float().cpu() - xa).abs().max() / xa.abs().max()).item() < 5e-2
    # measured logit error of the fp8 model along the oracle's history: every differing id must sit at a step whose oracle
    # margin is <= 2 x that error; the ceiling is the bf16 one (the weights ARE the oracle's: only activations are rounded)
    err, rep = check_low_precision_decode(m8, f8, ref, init, always, first, sp.eot, f"fp8 weights, {name}")
    print(f"\nfp8 weights [{name}]: max logit error vs the f32 oracle on the dequantised weights {rep['max_logit_err']:.4f} = "
          f"{rep['rel_err']:.4f} of the logit std, token match {rep['token_match']:.3f}")
    # a weight update or a dtype change invalidates the codes
    m8.load_weights({"decoder.ln.bias": Wdq["decoder.ln.bias"] + 1}, strict=False)
    assert m8.weights_format == "bfloat16"
    m8.quantize_weights()
    assert m8.weights_format == "fp8_e4m3"
    m8.set_dtype(torch.float32)
    assert m8.weights_format == "float32" and m8.packed()["cfg"].dec_w_dtype == 0


# ---------------------------------------------------------------------------------------------------------------------
# round 3: the HIP path against the WIDE golden fixtures (tests/golden/wide_model.npz: whisper-small width and large-v3
# dims -- 128 mels, 51 866 tokens, 100 languages -- from the transformers stand-in), not only against the oracle.
WIDE = {"small1": (R.ModelDimensions(80, 1500, 768, 12, 1, 51865, 448, 768, 12, 1), 41, 99),
        "large1": (R.ModelDimensions(128, 1500, 1280, 20, 1, 51866, 448, 1280, 20, 1), 43, 100)}


@pytest.mark.parametrize("name", sorted(WIDE))
def test_wide_golden_features_logits_loss_gradients_language_and_fp16_features(name, golden_dir, f32_mode):
    """float32 HIP path vs the stand-in's vectors: log-mel -> encoder slices, teacher-forced logits slices, masked-CE loss,
    slices of eight decoder gradients (DecoderTrainer.loss_and_grads vs the stand-in's autograd), detect_language at 99 / 100
    languages, greedy ids on f32 and on fp16-rounded features (DecodingOptions.fp16), and the fp16-rounded
    sinusoid table (App. C.3, Whisper(sinusoid_rounding="fp16")).  north_star tolerance: 1e-3 in fp32."""
    import whisper_ipa_amd as wipa
    from whisper_ipa_amd import ops
    from whisper_ipa_amd.decoding import detect_language, greedy_decode_tokens
    from whisper_ipa_amd.tokenizer import get_tokenizer
    from whisper_ipa_amd.training import DecoderTrainer

    g = np.load(os.path.join(golden_dir, "wide_model.npz"))
    dims, seed, n_lang = WIDE[name]
    split = f32_mode == "split"
    W = R.synthetic_weights(dims, seed=seed)
    clips = np.stack([R.synthetic_clip(2, 30.0), R.synthetic_clip(3, 7.0)])
    m = _model(dims, W, torch.float32, f32_split=split)
    mel = wipa.log_mel_spectrogram(clips, n_mels=dims.n_mels)
    assert abs(float(mel.abs().sum()) - g[f"{name}_mel_checksum"][0]) / g[f"{name}_mel_checksum"][0] < 1e-5
    feats = m.encoder(mel)
    rows, cols = g[f"{name}_enc_rows"], g[f"{name}_enc_cols"]
    assert np.abs(feats.cpu().numpy()[:, rows][:, :, cols] - g[f"{name}_enc_slices"]).max() < 1e-3
    tokens = torch.from_numpy(g[f"{name}_tokens"])
    logits = m.logits(tokens[:, :-1].cuda(), feats)
    lc = g[f"{name}_logit_cols"]
    assert np.abs(logits.cpu().numpy()[:, :, lc] - g[f"{name}_logit_slices"]).max() < 2e-3
    sp = R.SpecialTokens.multilingual(n_lang)
    tr = DecoderTrainer(m)
    loss, _, n_valid = tr.loss_and_grads(feats, tokens.cuda(), sp.eot)
    assert abs(float(loss) - g[f"{name}_loss"][0]) < 1e-3
    tok_rows = g[f"{name}_grad_tok_rows"]
    for key in [k for k in g.files if k.startswith(f"{name}_grad__")]:
        pname = key.split("__", 1)[1].replace("__", ".")
        want = g[key]
        got = tr.g(pname).cpu()
        if pname == "decoder.token_embedding.weight":
            got = got[tok_rows][:, :16]
        elif got.dim() == 1:
            got = got[: want.shape[0]]
        else:
            got = got[: want.shape[0], : want.shape[1]]
        assert float(np.abs(got.numpy() - want).max()) / (float(np.abs(want).max()) + 1e-12) < 3e-3, pname
        norm = float(g[f"{name}_gradnorm__{key.split('__', 1)[1]}"][0])
        assert abs(float(tr.g(pname).norm()) - norm) / norm < 2e-3, pname
    tok = get_tokenizer(True, num_languages=n_lang)
    lang, _ = detect_language(m, feats, tok)
    assert lang.tolist() == g[f"{name}_lang_tokens"].tolist()
    always, first = R.suppress_lists(sp)
    init = list(sp.sot_sequence_including_notimestamps(0))
    for tag, f in (("f32", feats), ("fp16feat", feats.half().float())):
        res = greedy_decode_tokens(m, f, init, always, first, -1, max_new_tokens=8, stop_on_eot=False)
        assert res.tokens.tolist() == g[f"{name}_greedy_{tag}_tokens"].tolist(), tag
    want_last = g[f"{name}_fp16feat_last_logit_slices"]  # the stand-in's FILTERED logits: -inf at suppressed ids
    ok = np.isfinite(want_last)
    assert ok.sum() >= 8 and np.abs(res.last_logits.cpu().numpy()[:, lc[:32]][ok] - want_last[ok]).max() < 3e-3
    # App. C.3: the fp16-rounded sinusoid table is an explicit switch, pinned on both sides
    from whisper_ipa_amd.whisper import ModelDimensions, Whisper
    m16 = Whisper(ModelDimensions(**dims.__dict__), dtype=torch.float32, f32_split=split, sinusoid_rounding="fp16")
    m16.load_weights(W)
    f16 = m16.encoder(mel)
    assert np.abs(f16.cpu().numpy()[:, rows][:, :, cols] - g[f"{name}_enc_fp16pos_slices"]).max() < 1e-3
    d = float((f16 - feats).abs().max())
    assert abs(d - float(g[f"{name}_enc_fp16pos_maxdiff"][0])) < 3e-4 and d > 2e-4


@pytest.mark.parametrize("name,dims", [
    ("small-width", R.ModelDimensions(80, 1500, 768, 12, 2, 51865, 448, 768, 12, 2)),
    ("large-v3-width", R.ModelDimensions(128, 1500, 1280, 20, 2, 51866, 448, 1280, 20, 2)),
])
def test_fp8_activation_encoder_on_the_fp8_mfma(name, dims):
    """BASELINE.json configs[4] "(CDNA4 fp8 MFMA)": quantize_weights(activations="fp8") runs the encoder's q|k, value, mlp1
    and mlp2 projections fp8 x fp8 on v_mfma_scale_f32_16x16x128_f8f6f4 (LayerNorm and GELU outputs quantised per row).
    Checked against (a) an EMULATION of exactly that arithmetic on the CPU oracle (same e4m3 rounding of the same
    activations, f32 products): features within bf16 tolerance; (b) the bf16-activation model on the same quantised weights
    -- the stand-in for "IPA-PER within 0.2 of bf16": feature difference, decode-step logit difference along the bf16
    model's ids, and the ids themselves, with the measured-error rule (a differing id only where the bf16 model's margin is
    <= 2 x the logit difference)."""
    import whisper_ipa_amd as wipa
    from parity_util import masked_margins
    from whisper_ipa_amd import ops
    from whisper_ipa_amd.decoding import forced_decode_logits, greedy_decode_tokens
    from whisper_ipa_amd.whisper import dequantize_fp8_e4m3, quantize_fp8_e4m3

    W = R.synthetic_weights(dims, seed=33)
    audio = np.stack([R.synthetic_clip(5, 30.0), R.synthetic_clip(6, 6.0)])
    mb = _model(dims, W, torch.bfloat16)
    mb.quantize_weights("fp8_e4m3")                      # bf16 activations on the quantised weights
    m8 = _model(dims, W, torch.bfloat16)
    m8.quantize_weights("fp8_e4m3", activations="fp8")   # fp8 activations too
    assert m8.activations_format == "fp8_e4m3" and m8.packed()["cfg"].enc_act_fp8 == 1 and mb.packed()["cfg"].enc_act_fp8 == 0
    mel = wipa.log_mel_spectrogram(audio, n_mels=dims.n_mels)
    ops.gemm_dispatch_counts(reset=True)
    f8 = m8.encoder(mel)
    assert ops.gemm_dispatch_counts(reset=True)["tile_fp8"] == 4 * dims.n_audio_layer
    fb = mb.encoder(mel)
    # (a) the same arithmetic emulated on the oracle: e4m3-round the LayerNorm outputs and the GELU output per row
    Wdq = {k: v.float().cpu() for k, v in m8.flat_parameters().items()}

    def q8(t):  # row-wise e4m3 with power-of-two scales, the product's rule
        shp = t.shape
        c, s = quantize_fp8_e4m3(t.reshape(-1, shp[-1]))
        return dequantize_fp8_e4m3(c, s).reshape(shp)

    def emulated_encoder(mel_t):
        import torch.nn.functional as F

        x = mel_t.transpose(1, 2)
        x = F.gelu(F.conv1d(x, Wdq["encoder.conv1.weight"].permute(0, 2, 1), Wdq["encoder.conv1.bias"], padding=1))
        x = F.gelu(F.conv1d(x, Wdq["encoder.conv2.weight"].permute(0, 2, 1), Wdq["encoder.conv2.bias"], stride=2, padding=1))
        x = x.transpose(1, 2) + R.sinusoids(dims.n_audio_ctx, dims.n_audio_state)
        H = dims.n_audio_head
        for i in range(dims.n_audio_layer):
            p = f"encoder.blocks.{i}"
            h = q8(R._layer_norm(x, Wdq, p + ".attn_ln"))
            q = F.linear(h, Wdq[p + ".attn.query.weight"], Wdq[p + ".attn.query.bias"])
            k = F.linear(h, Wdq[p + ".attn.key.weight"])
            v = F.linear(h, Wdq[p + ".attn.value.weight"], Wdq[p + ".attn.value.bias"])
            B_, T_, D_ = q.shape
            sc = (D_ // H) ** -0.25
            qh = q.view(B_, T_, H, -1).permute(0, 2, 1, 3) * sc
            kh = k.view(B_, T_, H, -1).permute(0, 2, 3, 1) * sc
            a = (torch.softmax(qh @ kh, -1) @ v.view(B_, T_, H, -1).permute(0, 2, 1, 3)).permute(0, 2, 1, 3).reshape(B_, T_, D_)
            x = x + F.linear(a, Wdq[p + ".attn.out.weight"], Wdq[p + ".attn.out.bias"])
            h = q8(R._layer_norm(x, Wdq, p + ".mlp_ln"))
            u = q8(F.gelu(F.linear(h, Wdq[p + ".mlp1.weight"], Wdq[p + ".mlp1.bias"])))
            x = x + F.linear(u, Wdq[p + ".mlp2.weight"], Wdq[p + ".mlp2.bias"])
        return R._layer_norm(x, Wdq, "encoder.ln_post")

    with torch.no_grad():
        mel_ref = np.stack([R.log_mel_spectrogram(x, dims.n_mels) for x in audio])
        xa8 = emulated_encoder(torch.from_numpy(mel_ref))
        xa = R.encoder_forward(Wdq, dims, torch.from_numpy(mel_ref))
    rel_emul = ((f8.float().cpu() - xa8).abs().max() / xa8.abs().max()).item()
    rms_emul = ((f8.float().cpu() - xa8).pow(2).mean().sqrt() / xa8.pow(2).mean().sqrt()).item()
    rel_vs_f32 = ((f8.float().cpu() - xa).abs().max() / xa.abs().max()).item()
    rms_vs_bf16 = ((f8.float() - fb.float()).pow(2).mean().sqrt() / fb.float().pow(2).mean().sqrt()).item()
    # (b) downstream: decode-step logits of the two models along the bf16-activation model's ids
    sp = R.SpecialTokens.multilingual(100 if dims.n_vocab == 51866 else 99)
    always, first = R.suppress_lists(sp)
    init = list(sp.sot_sequence_including_notimestamps(0))
    rb = greedy_decode_tokens(mb, fb, init, always, first, sp.eot, max_new_tokens=16, stop_on_eot=False)
    tb, cb = forced_decode_logits(mb, fb, rb.tokens, 4, always, first, sp.eot)
    t8, c8 = forced_decode_logits(m8, f8, rb.tokens, 4, always, first, sp.eot)
    keep = torch.ones(dims.n_vocab, dtype=torch.bool, device=tb.device)
    keep[list(always)] = False
    err = (t8 - tb)[:, :, keep].abs().amax(dim=-1).cpu().numpy()
    spread = float(tb[:, :, keep].std())
    margins = masked_margins(tb, always, first)
    flips = c8 != rb.tokens[:, 4:]
    print(f"\nfp8 activations [{name}]: features vs the emulated fp8 arithmetic {rel_emul:.3e} max / {rms_emul:.3e} rms, vs the f32 oracle {rel_vs_f32:.3e}, "
          f"vs the bf16-activation model rms {rms_vs_bf16:.3e}; decode logits differ by at most {err.max():.4f} = {err.max() / spread:.4f} of the "
          f"logit std; {int(flips.sum())} of {flips.size} teacher-forced choices differ (largest margin among them "
          f"{margins[flips].max() if flips.any() else 0.0:.4f})")
    # The emulation cannot be CLOSER to the kernels than the quantisation noise itself: the GPU quantises LayerNorm outputs
    # computed from a bf16-path residual stream, the emulation from an f32 one; a 0.4 % input difference straddles an e4m3
    # rounding boundary (6-12 % steps) for roughly one element in ten, and each such flip is a whole quantum -- as much rms
    # as the rounding noise of all elements together.  What it shows is that the kernels' error has the size this arithmetic
    # has by construction (the exactness of the GEMM and of the quantisers is pinned in tests/test_gpu_kernels.py).
    assert rms_emul < 1.5 * rms_vs_bf16 and rms_emul < 0.08 and rel_emul < 0.15, (rms_emul, rms_vs_bf16, rel_emul)
    assert rms_vs_bf16 < 0.15, rms_vs_bf16              # e4m3 activations: a few percent of the feature rms
    assert (margins[flips] <= 2.0 * err[flips]).all()   # ids differ only where the measured logit difference allows it
    assert err.max() < 0.35 * spread, (err.max(), spread)


# ---- several batches in flight: whisper_ipa_amd.pipeline (the schedule bench.py times and the scripts call) ----------------------

def test_transcribe_batches_early_stop_language_detection_and_order_match_decode():
    """pipeline.transcribe_batches against ``decode`` one batch at a time (float32, a model that does end its rows): ragged
    batch sizes, more batches than slots, early stop armed (chunks of 3 steps, EOT probes behind them), per-clip language
    detection on the device.  Every field the reference's callers read (scripts/evaluate_model.py:190-201,
    scripts/train_whisper_ipa.py:352-362) is the one ``decode`` gives: tokens, text, language, avg_logprob; results come
    back in input order; the model's cross_splits setting is restored."""
    import whisper_ipa_amd as wipa

    dims = R.ModelDimensions(80, 1500, 64, 1, 1, 51865, 448, 64, 1, 1)
    W = R.synthetic_weights(dims, seed=3)
    m = _model(dims, W, torch.float32)
    sp = R.SpecialTokens.multilingual()
    allowed = {sp.eot, 100, 200, 300}
    suppress = [t for t in range(dims.n_vocab) if t not in allowed]
    torch.manual_seed(1)
    batches = [torch.randn(b, 1500, 64) for b in (3, 1, 4, 2, 3, 4, 1)]
    for language in ("en", None):
        opts = wipa.DecodingOptions(language=language, without_timestamps=True, fp16=False, suppress_tokens=suppress, suppress_blank=False,
                                    sample_len=40)
        want = [wipa.decode(m, b.cuda(), opts) for b in batches]
        before = m.cross_splits
        got = list(wipa.transcribe_batches(m, batches, opts, passes_in_flight=3, check_every=3))
        assert m.cross_splits == before
        assert [r.index for r in got] == list(range(len(batches)))
        lens = set()
        for r, w in zip(got, want):
            assert len(r.results) == len(w)
            for a, b in zip(r.results, w):
                assert a.tokens == b.tokens and a.text == b.text and a.language == b.language
                assert abs(a.avg_logprob - b.avg_logprob) < 1e-4
                if language is None:
                    assert max(abs(a.language_probs[k] - b.language_probs[k]) for k in b.language_probs) < 1e-5
            lens.add(r.n_steps)
        assert len(lens) > 1 and min(lens) < 40, lens  # the passes did stop early, at different lengths
    # decode groups (G consecutive batches decode as one chain of rows): every batch still gets the result it gets alone --
    # its own early-stop length included -- with full and partial groups, with and without language detection
    for language in ("en", None):
        opts = wipa.DecodingOptions(language=language, without_timestamps=True, fp16=False, suppress_tokens=suppress, suppress_blank=False,
                                    sample_len=40)
        want = [wipa.decode(m, b.cuda(), opts) for b in batches]
        for G, P in ((2, 2), (3, 1), (4, 2)):
            got = list(wipa.transcribe_batches(m, batches, opts, passes_in_flight=P, check_every=3, decode_group=G))
            assert [r.index for r in got] == list(range(len(batches))), (G, P)
            for r, w in zip(got, want):
                assert [x.tokens for x in r.results] == [x.tokens for x in w] and [x.language for x in r.results] == [x.language for x in w]
                assert max(abs(a.avg_logprob - b.avg_logprob) for a, b in zip(r.results, w)) < 1e-4
                alone = [len(x.tokens) for x in w]
                assert r.n_steps <= 40 and (r.n_steps == 40 or r.n_steps == max(alone) + 1), (G, P, r.n_steps, alone)
    # the serial schedule and a prefetching iterator give the same
    opts = wipa.DecodingOptions(language="en", without_timestamps=True, fp16=False, suppress_tokens=suppress, suppress_blank=False, sample_len=40)
    a = [r.rows() for r in wipa.transcribe_batches(m, batches, opts, passes_in_flight=1)]
    b = [r.rows() for r in wipa.transcribe_batches(m, iter(batches), opts, passes_in_flight=4, prefetch=2)]
    assert a == b == [[r.tokens for r in wipa.decode(m, x.cuda(), opts)] for x in batches]
    with pytest.raises(wipa._lib.WipaError):
        list(wipa.transcribe_batches(m, [torch.zeros(1, 2, 3, 4)], opts))  # neither audio, mel nor features


def test_transcribe_batches_320_clips_ids_identical_to_the_serial_path():
    """the 4-passes-in-flight schedule at FULL size (whisper-small 12+12 bf16, 64-clip batches of distinct
    clips, 5 batches = 320 clips so a stream set is reused) gives every clip the ids of the serial path -- one batch at a
    time through log_mel_padded -> encode_padded -> greedy_decode_tokens in the same streaming-launch setting (cross_splits = 2,
    what the schedule sets) -- bit for bit; with early stop armed (the random-init model never ends a row) the same again;
    audio handed over on the host the same again.  Informational: the match rate against the serial path in the library's
    default setting (4 splits), where only the order of the softmax merges differs."""
    import bench
    from whisper_ipa_amd import audio as A
    from whisper_ipa_amd.decoding import DecodingOptions, greedy_decode_tokens
    from whisper_ipa_amd.pipeline import PIPELINE_CROSS_SPLITS, transcribe_batches
    from whisper_ipa_amd.whisper import Whisper

    dims, W = bench.synthetic_weights_small(0)
    m = Whisper(dims, dtype=torch.bfloat16)
    m.load_weights(W)
    del W
    init, always, first, eot = bench.decode_setup()
    opts = DecodingOptions(language="en", without_timestamps=True)
    n_new = 24
    host = [torch.from_numpy(bench.synthetic_audio(64 * i, 64)) for i in range(5)]
    dev = [a.cuda() for a in host]

    def serial(splits):
        m.cross_splits = splits
        out = []
        for a in dev:
            feats = m.encode_padded(A.log_mel_padded(a, dims.n_mels, torch.bfloat16), 64)
            out.append(greedy_decode_tokens(m, feats, init, always, first, eot, max_new_tokens=n_new, stop_on_eot=False).tokens)
        m.cross_splits = 0
        return out

    want = serial(PIPELINE_CROSS_SPLITS)
    got = [r.tokens for r in transcribe_batches(m, dev, opts, passes_in_flight=4, max_new_tokens=n_new, stop_on_eot=False)]
    assert m.cross_splits == 0
    assert len(got) == 5 and all((g == w).all() for g, w in zip(got, want))
    # the equality above is not vacuous: the rows are not all alike.  A random-init (std 0.02) model depends only weakly on its
    # audio -- measured r05: 33 distinct id rows among the 320 clips (test_full_size_bench_workload_properties asks > 8 of 64)
    distinct = len({tuple(r) for t in got for r in t[:, 4:].tolist()})
    print(f"\ndistinct id rows among the 320 clips: {distinct}")
    assert distinct > 16, distinct
    armed = [r.tokens for r in transcribe_batches(m, dev, opts, passes_in_flight=4, max_new_tokens=n_new, check_every=8)]
    assert all((g == w).all() for g, w in zip(armed, want))
    from_host = [r.tokens for r in transcribe_batches(m, host[:2], opts, passes_in_flight=2, max_new_tokens=n_new, stop_on_eot=False)]
    assert all((g == w).all() for g, w in zip(from_host, want))
    grouped = [r.tokens for r in transcribe_batches(m, dev, opts, passes_in_flight=2, max_new_tokens=n_new, stop_on_eot=False, decode_group=2)]
    assert len(grouped) == 5 and all((g == w).all() for g, w in zip(grouped, want))  # 128-row chains (and a last group of one batch)
    one = [r.tokens for r in transcribe_batches(m, dev[:2], opts, passes_in_flight=1, max_new_tokens=n_new, stop_on_eot=False)]
    dflt = serial(0)
    assert all((g == w).all() for g, w in zip(one, dflt))  # one batch at a time: the model's own setting
    match = np.mean([(g[:, 4:] == w[:, 4:]).mean() for g, w in zip(got, dflt)])
    print(f"\n4 passes in flight (2 frame splits) vs serial default (4 splits), random-init bf16: token match {match:.4f}")


@pytest.mark.parametrize("form", ["cached", "fp8-weights", "f32"])
def test_transcribe_batches_other_model_forms_match_the_serial_path(small2, form):
    """The forms `bench.py`'s other_configs drive through the schedule besides absorbed bf16: mlx_whisper's cached K / V
    (what whisper-large-v3's 20 heads take), fp8 e4m3 weights, and a float32 model -- 3 passes in flight over 5 ragged batches give
    every clip the ids of one batch at a time through greedy_decode_tokens on the same features, bit for bit."""
    from whisper_ipa_amd.decoding import DecodingOptions, greedy_decode_tokens
    from whisper_ipa_amd.pipeline import transcribe_batches

    W, mels, xa = small2
    sp = R.SpecialTokens.multilingual()
    always, first = R.suppress_lists(sp)
    init = list(sp.sot_sequence_including_notimestamps(0))
    dtype = torch.float32 if form == "f32" else torch.bfloat16
    m = _model(SMALL2, W, dtype, cross_attention="cached" if form == "cached" else "auto")
    if form == "fp8-weights":
        m.quantize_weights("fp8_e4m3")
    torch.manual_seed(5)
    base = xa.cuda().to(dtype)
    batches = [(base[torch.randint(0, 2, (b,))] + 0.05 * torch.randn(b, 1500, SMALL2.n_audio_state, device="cuda").to(dtype)).contiguous()
               for b in (3, 5, 2, 4, 3)]
    opts = DecodingOptions(language="en", without_timestamps=True, fp16=False)
    want = [greedy_decode_tokens(m, f, init, always, first, sp.eot, max_new_tokens=12, stop_on_eot=False).tokens for f in batches]
    got = [r.tokens for r in transcribe_batches(m, batches, opts, passes_in_flight=3, max_new_tokens=12, stop_on_eot=False,
                                                cross_splits=m.cross_splits)]
    assert len(got) == 5 and all(g.shape == w.shape and (g == w).all() for g, w in zip(got, want)), form
    # not vacuous: the batches are drawn from the fixture's TWO clips (the 0.05 noise does not move their greedy ids: measured 2-3
    # distinct rows), and both clips' ids appear
    assert len({tuple(r) for g in got for r in g[:, 4:].tolist()}) >= 2
