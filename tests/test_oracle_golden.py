"""Pin oracle/whisper_ref.py against the committed golden fixtures
(tests/golden/*.npz, made by tools/make_golden.py from the transformers
stand-in).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import whisper_ref as R
from parity_util import norm64

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


def test_clip_grad_dict_walks_dicts_only_as_the_reference_wrote_it():
    """reference scripts/train_whisper_ipa.py:287-303: ``isinstance(value, dict)`` recurses, ``hasattr(value, 'shape')`` clips,
    everything else -- a LIST included -- is returned as it came.  mlx_whisper keeps ``decoder.blocks`` in a list (the
    reference's flatten_params :50-53 has the list branch, and its checkpoint keys are ``decoder.blocks.{i}.``), so no block
    tensor is ever clipped; the four tensors outside the list are."""
    big = lambda *shape: torch.full(shape, 3.0)
    flat = {
        "decoder.token_embedding.weight": big(4, 4), "decoder.positional_embedding": big(2, 4),
        "decoder.blocks.0.attn.query.weight": big(4, 4), "decoder.blocks.0.mlp1.bias": big(8),
        "decoder.blocks.1.attn.query.weight": big(4, 4), "decoder.blocks.1.mlp_ln.weight": big(4) * 0.01,
        "decoder.ln.weight": big(4), "decoder.ln.bias": big(4) * 0.01,
    }
    tree = R.unflatten_params(flat)
    assert isinstance(tree["decoder"]["blocks"], list) and len(tree["decoder"]["blocks"]) == 2
    assert isinstance(tree["decoder"]["blocks"][0], dict) and isinstance(tree["decoder"]["ln"], dict)
    assert list(R.flatten_params(tree)) == list(flat) and all(R.flatten_params(tree)[k] is flat[k] for k in flat)
    walked = R.clip_grad_dict(tree, 1.0)
    assert walked["decoder"]["blocks"] is tree["decoder"]["blocks"]  # :299-300: the very same list object comes back
    out = R.flatten_params(walked)
    for k, g in flat.items():
        n0, n1 = norm64(g), norm64(out[k])
        if ".blocks." in k:
            assert torch.equal(out[k], g) and not R.clipped_by_reference(k)
        elif n0 > 1:
            assert abs(n1 - n0 / (n0 + 1e-6)) < 1e-6 and R.clipped_by_reference(k)
        else:
            assert torch.equal(out[k], g)
    assert R.clip_gradients(flat, 1.0, "reference").keys() == flat.keys()
    every = R.clip_gradients(flat, 1.0, "all")
    assert all(norm64(every[k]) <= 1.0 + 1e-7 for k in flat)
    # the product marks the same tensors (what wipa_clip_adamw's seg_clip carries)
    from whisper_ipa_amd.training import clip_reaches

    assert all(clip_reaches(k, "reference") == R.clipped_by_reference(k) and clip_reaches(k, "all") for k in flat)


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


# ---------------------------------------------------------------------------------------------------------------------
# round 3: the oracle beyond MICRO dims (tests/golden/wide_model.npz, tools/make_golden.py golden_wide): whisper-small
# width and whisper-large-v3 dims (128 mels, 51 866 tokens, 100 languages), one encoder + one decoder layer, with a
# GRADIENT fixture from the stand-in's autograd, the fp16-features decode path and the fp16-rounded
# sinusoid table (App. C.3).
SMALL1 = R.ModelDimensions(80, 1500, 768, 12, 1, 51865, 448, 768, 12, 1)
LARGE1 = R.ModelDimensions(128, 1500, 1280, 20, 1, 51866, 448, 1280, 20, 1)
WIDE = {"small1": (SMALL1, 41, 99), "large1": (LARGE1, 43, 100)}


@pytest.fixture(scope="module")
def gwide(golden_dir):
    return np.load(os.path.join(golden_dir, "wide_model.npz"))


@pytest.fixture(scope="module", params=sorted(WIDE))
def wide(request):
    name = request.param
    dims, seed, n_lang = WIDE[name]
    W = R.synthetic_weights(dims, seed=seed)
    mels = np.stack([R.log_mel_spectrogram(R.synthetic_clip(2, 30.0), dims.n_mels), R.log_mel_spectrogram(R.synthetic_clip(3, 7.0), dims.n_mels)])
    with torch.no_grad():
        xa = R.encoder_forward(W, dims, torch.from_numpy(mels))
    return name, dims, W, mels, xa, R.SpecialTokens.multilingual(n_lang)


def test_wide_encoder_matches_standin(gwide, wide):
    name, dims, W, mels, xa, sp = wide
    assert abs(float(np.abs(mels).sum()) - gwide[f"{name}_mel_checksum"][0]) / gwide[f"{name}_mel_checksum"][0] < 1e-6
    rows, cols = gwide[f"{name}_enc_rows"], gwide[f"{name}_enc_cols"]
    np.testing.assert_allclose(xa[:, rows][:, :, cols].numpy(), gwide[f"{name}_enc_slices"], atol=3e-4)
    st = gwide[f"{name}_enc_stats"]
    assert abs(xa.mean().item() - st[0]) < 1e-4 and abs(xa.std().item() - st[1]) < 1e-4 and abs(xa.abs().sum().item() - st[2]) / st[2] < 1e-5


def test_wide_logits_loss_and_decoder_gradients_match_standin(gwide, wide):
    """teacher-forced logits, masked CE and slices of eight decoder gradients (tied embedding, positional table, self
    query, cross key, cross value bias, mlp1, a LayerNorm weight, the final LayerNorm bias) against the stand-in's autograd:
    what scripts/train_whisper_ipa.py:223-263,284 compute, at whisper-small width and at large-v3's vocabulary / mel count."""
    name, dims, W, mels, xa, sp = wide
    tokens = torch.from_numpy(gwide[f"{name}_tokens"])
    names = [k.split("__", 1)[1].replace("__", ".") for k in gwide.files if k.startswith(f"{name}_grad__")]
    assert len(names) == 8
    leaves = {k: W[k].detach().clone().requires_grad_(True) for k in names}
    Wl = dict(W)
    Wl.update(leaves)
    logits = R.decoder_forward(Wl, dims, tokens[:, :-1], xa)
    cols = gwide[f"{name}_logit_cols"]
    np.testing.assert_allclose(logits[:, :, cols].detach().numpy(), gwide[f"{name}_logit_slices"], atol=1e-3)
    st = gwide[f"{name}_logit_stats"]
    assert abs(logits.mean().item() - st[0]) < 1e-4 and abs(logits.std().item() - st[1]) < 1e-3
    loss = R.loss_from_features(Wl, dims, xa, tokens, sp.eot)
    assert abs(loss.item() - gwide[f"{name}_loss"][0]) < 1e-4
    grads = dict(zip(names, torch.autograd.grad(loss, [leaves[k] for k in names])))
    tok_rows = gwide[f"{name}_grad_tok_rows"]
    for k, g in grads.items():
        key = k.replace(".", "__")
        want = gwide[f"{name}_grad__{key}"]
        if k == "decoder.token_embedding.weight":
            got = g[tok_rows][:, :16]
        elif g.dim() == 1:
            got = g[: want.shape[0]]
        else:
            got = g[: want.shape[0], : want.shape[1]]
        scale = float(np.abs(want).max()) + 1e-12
        assert float(np.abs(got.numpy() - want).max()) / scale < 2e-3, k
        norm = float(gwide[f"{name}_gradnorm__{key}"][0])
        assert abs(norm64(g) - norm) / norm < 1e-3, k


def test_wide_language_detection_and_greedy_with_fp16_features(gwide, wide):
    """detect_language at 99 / 100 languages, greedy ids on f32 features and on fp16-ROUNDED features (App. C.2: the
    DecodingOptions.fp16=True default of transcribe_single.py:49-52) against the cache-free stand-in loop."""
    name, dims, W, mels, xa, sp = wide
    with torch.no_grad():
        assert R.detect_language(W, dims, xa, sp).tolist() == gwide[f"{name}_lang_tokens"].tolist()
        always, first = R.suppress_lists(sp)
        init = sp.sot_sequence_including_notimestamps(0)
        for tag, fp16 in (("f32", False), ("fp16feat", True)):
            want, margins = gwide[f"{name}_greedy_{tag}_tokens"], gwide[f"{name}_greedy_{tag}_margins"]
            res = R.greedy_decode(W, dims, xa, init, always, first, -1, sample_len=8, stop_on_eot=False, fp16_features=fp16, keep_logits=True)
            assert margins.min() > 1e-3
            assert res.tokens.tolist() == want.tolist(), tag
            np.testing.assert_allclose(res.margins, margins, atol=2e-3)
            if fp16:
                cols = gwide[f"{name}_logit_cols"][:32]
                np.testing.assert_allclose(res.step_logits[:, -1][:, cols], gwide[f"{name}_fp16feat_last_logit_slices"], atol=2e-3)


def test_wide_sinusoid_table_choice_is_explicit(gwide, wide):
    """mlx_whisper builds AudioEncoder._positional_embedding in the load dtype (fp16) and
    set_dtype(float32) leaves it alone [UPSTREAM-UNVERIFIED].  The oracle's DEFAULT is the f32 table (what openai/whisper
    computes); the fp16-rounded table is the explicit override W["encoder._positional_embedding"] -- both pinned here, and
    the product follows the same switch (Whisper(..., sinusoid_rounding="fp16"), tests/test_gpu_model.py)."""
    name, dims, W, mels, xa, sp = wide
    rows, cols = gwide[f"{name}_enc_rows"], gwide[f"{name}_enc_cols"]
    W16 = dict(W)
    W16["encoder._positional_embedding"] = R.sinusoids(dims.n_audio_ctx, dims.n_audio_state).half().float()
    with torch.no_grad():
        xa16 = R.encoder_forward(W16, dims, torch.from_numpy(mels))
    np.testing.assert_allclose(xa16[:, rows][:, :, cols].numpy(), gwide[f"{name}_enc_fp16pos_slices"], atol=3e-4)
    d = float((xa16 - xa).abs().max())
    assert abs(d - float(gwide[f"{name}_enc_fp16pos_maxdiff"][0])) < 2e-4 and d > 2e-4  # a visible, but sub-1e-3 shift
