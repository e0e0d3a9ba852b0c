"""Pin oracle/whisper_ref.py against the committed golden fixtures
(tests/golden/*.npz, made by tools/make_golden.py from the transformers
stand-in).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import whisper_ref as R

MICRO = R.ModelDimensions(80, 1500, 128, 2, 2, 51865, 448, 128, 2, 2)


@pytest.fixture(scope="module")
def gmel(golden_dir):
    return np.load(os.path.join(golden_dir, "mel.npz"))


@pytest.fixture(scope="module")
def gmodel(golden_dir):
    return np.load(os.path.join(golden_dir, "micro_model.npz"))


@pytest.fixture(scope="module")
def micro():
    W = R.synthetic_weights(MICRO, seed=7)
    mels = np.stack([R.log_mel_spectrogram(R.synthetic_clip(0, 30.0)), R.log_mel_spectrogram(R.synthetic_clip(1, 5.0))])
    with torch.no_grad():
        xa = R.encoder_forward(W, MICRO, torch.from_numpy(mels))
    return W, mels, xa


@pytest.mark.parametrize("n_mels", [80, 128])
def test_mel_filters_match_standin(gmel, n_mels):
    f = R.mel_filters(n_mels)
    assert f.shape == (n_mels, 201)
    np.testing.assert_allclose(f, gmel[f"filters_{n_mels}"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("n_mels", [80, 128])
@pytest.mark.parametrize("name,idx,secs", [("full", 0, 30.0), ("short", 1, 5.0)])
def test_log_mel_matches_standin(gmel, n_mels, name, idx, secs):
    mel = R.log_mel_spectrogram(R.synthetic_clip(idx, secs), n_mels)
    assert mel.shape == (3000, n_mels) and mel.dtype == np.float32
    rows = gmel[f"{name}_{n_mels}_rows"]
    np.testing.assert_allclose(mel[rows], gmel[f"{name}_{n_mels}_slices"], atol=2e-4)
    st = gmel[f"{name}_{n_mels}_stats"]
    assert abs(mel.mean() - st[0]) < 1e-4 and abs(mel.std() - st[1]) < 1e-4
    assert abs(np.abs(mel).sum() - st[2]) / st[2] < 1e-4
    np.testing.assert_allclose(mel.mean(axis=0), gmel[f"{name}_{n_mels}_colmean"], atol=1e-4)


def test_pad_or_trim():
    a = np.arange(10, dtype=np.float32)
    assert R.pad_or_trim(a, 4).tolist() == [0, 1, 2, 3]
    out = R.pad_or_trim(a, 12)
    assert out.shape == (12,) and out[10:].tolist() == [0, 0] and out[:10].tolist() == a.tolist()
    assert R.pad_or_trim(np.zeros((2, 5)), 7).shape == (2, 7)


def test_encoder_matches_standin(gmodel, micro):
    W, mels, xa = micro
    assert abs(float(np.abs(mels).sum()) - gmodel["mel_checksum"][0]) / gmodel["mel_checksum"][0] < 1e-6
    rows = gmodel["enc_rows"]
    np.testing.assert_allclose(xa[:, rows].numpy(), gmodel["enc_slices"], atol=2e-4)
    st = gmodel["enc_stats"]
    assert abs(xa.mean().item() - st[0]) < 1e-4 and abs(xa.std().item() - st[1]) < 1e-4


def test_logits_and_loss_match_standin(gmodel, micro):
    W, mels, xa = micro
    tokens = torch.from_numpy(gmodel["tokens"])
    with torch.no_grad():
        logits = R.decoder_forward(W, MICRO, tokens[:, :-1], xa)
        loss = R.loss_from_features(W, MICRO, xa, tokens, 50257)
    cols = gmodel["logit_cols"]
    np.testing.assert_allclose(logits[:, :, cols].numpy(), gmodel["logit_slices"], atol=1e-3)
    assert abs(loss.item() - gmodel["loss"][0]) < 1e-4
    mask = R.loss_mask(tokens[:, 1:], 50257).numpy()
    assert (mask == gmodel["loss_mask"]).all()
    # first EOT is kept, later (padding) EOTs are dropped (train_whisper_ipa.py:242-247)
    assert mask[1].sum() == 3 + 5 + 1 and mask[0].all()


def test_greedy_kv_cache_matches_uncached_standin(gmodel, micro):
    W, mels, xa = micro
    sp = R.SpecialTokens.multilingual()
    always, first = R.suppress_lists(sp)
    gold = gmodel["greedy_tokens"]
    n = gold.shape[1] - 4
    with torch.no_grad():
        r = R.greedy_decode(W, MICRO, xa, sp.sot_sequence_including_notimestamps(0), always, first, sp.eot,
                            sample_len=n, stop_on_eot=False)
    assert r.tokens.shape == gold.shape
    assert (r.tokens == gold).all()
    np.testing.assert_allclose(r.margins, gmodel["greedy_margins"], atol=2e-3)


def test_detect_language_matches_standin(gmodel, micro):
    W, mels, xa = micro
    sp = R.SpecialTokens.multilingual()
    with torch.no_grad():
        lang = R.detect_language(W, MICRO, xa, sp)
    assert (lang == gmodel["lang_tokens"]).all()


def test_special_tokens_known_answers():
    # WHISPER_IPA_RESEARCH_STANDALONE.md:333-338
    sp = R.SpecialTokens.multilingual()
    assert (sp.eot, sp.sot, sp.lang_first, sp.transcribe, sp.no_timestamps) == (50257, 50258, 50259, 50359, 50363)
    assert sp.sot_sequence_including_notimestamps(0) == (50258, 50259, 50359, 50363)
    always, first = R.suppress_lists(sp)
    assert first == [220, 50257]
    for t in (50258, 50358, 50359, 50360, 50361, 50362):
        assert t in always
    assert 50257 not in always and 50363 not in always


def test_eot_latch_and_stop():
    """Rows that emitted EOT keep emitting EOT; loop stops when all rows ended."""
    torch.manual_seed(0)
    dims = R.ModelDimensions(80, 1500, 64, 1, 1, 51865, 448, 64, 1, 1)
    W = R.synthetic_weights(dims, seed=3)
    xa = torch.randn(3, 1500, 64)
    sp = R.SpecialTokens.multilingual()
    allowed = {sp.eot, 100, 200, 300}
    always = [t for t in range(dims.n_vocab) if t not in allowed]
    with torch.no_grad():
        r = R.greedy_decode(W, dims, xa, sp.sot_sequence_including_notimestamps(0), always, [], sp.eot, sample_len=40)
    body = r.tokens[:, 4:]
    assert set(np.unique(body)).issubset(allowed)
    for row in body:
        idx = np.where(row == sp.eot)[0]
        if len(idx):
            assert (row[idx[0]:] == sp.eot).all()
    assert (body[:, -1] == sp.eot).all() or r.n_steps == 40


def test_clip_and_adamw_semantics():
    g = torch.tensor([3.0, 4.0])
    c = R.clip_per_tensor(g, 1.0)
    assert abs(torch.linalg.norm(c).item() - 5.0 / (5.0 + 1e-6)) < 1e-6
    small = torch.tensor([0.3, 0.4])
    assert torch.equal(R.clip_per_tensor(small, 1.0), small)
    p = torch.tensor([1.0])
    p2, m, v = R.adamw_mlx(p, torch.tensor([0.5]), torch.zeros(1), torch.zeros(1), lr=1e-2)
    m_e, v_e = 0.05, 0.001 * 0.25
    exp = 1.0 * (1 - 1e-2 * 0.01) - 1e-2 * m_e / (np.sqrt(v_e) + 1e-8)
    assert abs(p2.item() - exp) < 1e-6 and abs(m.item() - m_e) < 1e-7 and abs(v.item() - v_e) < 1e-9


def test_train_step_reduces_loss():
    torch.manual_seed(0)
    dims = R.ModelDimensions(80, 1500, 64, 1, 1, 51865, 448, 64, 1, 1)
    W = R.synthetic_weights(dims, seed=1)
    xa = torch.randn(2, 1500, 64)
    tokens = torch.tensor([[50258, 50259, 50359, 50363, 11, 12, 13, 50257], [50258, 50259, 50359, 50363, 21, 50257, 50257, 50257]])
    names = [k for k in W if k.startswith("decoder.")]
    state = {}
    losses = []
    for _ in range(3):
        leaves = {k: W[k].detach().clone().requires_grad_(True) for k in names}
        Wl = dict(W)
        Wl.update(leaves)
        loss = R.loss_from_features(Wl, dims, xa, tokens, 50257)
        grads = torch.autograd.grad(loss, [leaves[k] for k in names])
        for k, g in zip(names, grads):
            g = R.clip_per_tensor(g)
            m, v = state.get(k, (torch.zeros_like(g), torch.zeros_like(g)))
            W[k], m, v = R.adamw_mlx(W[k], g, m, v, lr=1e-3)
            state[k] = (m, v)
        losses.append(loss.item())
    assert losses[-1] < losses[0]
