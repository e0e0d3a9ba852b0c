#!/usr/bin/env python3
"""Generate tests/golden/*.npz from a STAND-IN of independent lineage.

The true reference arithmetic (mlx_whisper 0.4.3 / mlx 0.30.0) is not in
/root/reference and cannot be installed here, and the reference holds no golden
vector for this path.  The fixtures therefore come from
the ``transformers`` Whisper classes that happen to be installed in the build
container (same architecture, different code lineage), on seeded synthetic
weights and audio.  They pin oracle/whisper_ref.py; they do not pin the oracle
against mlx_whisper itself ("parity unpinned").

Run from the repo root:  python tools/make_golden.py
Needs: transformers, torch (CPU).  Never imported by the product or the tests.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import whisper_ref as R  # noqa: E402  (only for seeds/weights/names)

OUT = os.path.join(ROOT, "tests", "golden")

MICRO = R.ModelDimensions(80, 1500, 128, 2, 2, 51865, 448, 128, 2, 2)


def hf_model(dims: R.ModelDimensions, W):
    from transformers.models.whisper.configuration_whisper import WhisperConfig
    from transformers.models.whisper.modeling_whisper import WhisperForConditionalGeneration

    cfg = WhisperConfig(
        vocab_size=dims.n_vocab, num_mel_bins=dims.n_mels, encoder_layers=dims.n_audio_layer,
        encoder_attention_heads=dims.n_audio_head, decoder_layers=dims.n_text_layer,
        decoder_attention_heads=dims.n_text_head, decoder_ffn_dim=4 * dims.n_text_state,
        encoder_ffn_dim=4 * dims.n_audio_state, d_model=dims.n_audio_state,
        max_source_positions=dims.n_audio_ctx, max_target_positions=dims.n_text_ctx,
        attn_implementation="eager",
    )
    m = WhisperForConditionalGeneration(cfg).eval().float()
    sd = {}

    def put(dst, src, t=None):
        v = W[src]
        sd[dst] = (t(v) if t else v).clone()

    put("model.encoder.conv1.weight", "encoder.conv1.weight", lambda v: v.permute(0, 2, 1))
    put("model.encoder.conv1.bias", "encoder.conv1.bias")
    put("model.encoder.conv2.weight", "encoder.conv2.weight", lambda v: v.permute(0, 2, 1))
    put("model.encoder.conv2.bias", "encoder.conv2.bias")
    sd["model.encoder.embed_positions.weight"] = R.sinusoids(dims.n_audio_ctx, dims.n_audio_state)

    def block(hf, mx, cross):
        amap = [("self_attn", "attn", "self_attn_layer_norm")]
        if cross:
            amap.append(("encoder_attn", "cross_attn", "encoder_attn_layer_norm"))
        for ha, ma, hln in amap:
            put(f"{hf}.{ha}.q_proj.weight", f"{mx}.{ma}.query.weight")
            put(f"{hf}.{ha}.q_proj.bias", f"{mx}.{ma}.query.bias")
            put(f"{hf}.{ha}.k_proj.weight", f"{mx}.{ma}.key.weight")
            put(f"{hf}.{ha}.v_proj.weight", f"{mx}.{ma}.value.weight")
            put(f"{hf}.{ha}.v_proj.bias", f"{mx}.{ma}.value.bias")
            put(f"{hf}.{ha}.out_proj.weight", f"{mx}.{ma}.out.weight")
            put(f"{hf}.{ha}.out_proj.bias", f"{mx}.{ma}.out.bias")
            put(f"{hf}.{hln}.weight", f"{mx}.{ma}_ln.weight")
            put(f"{hf}.{hln}.bias", f"{mx}.{ma}_ln.bias")
        put(f"{hf}.fc1.weight", f"{mx}.mlp1.weight")
        put(f"{hf}.fc1.bias", f"{mx}.mlp1.bias")
        put(f"{hf}.fc2.weight", f"{mx}.mlp2.weight")
        put(f"{hf}.fc2.bias", f"{mx}.mlp2.bias")
        put(f"{hf}.final_layer_norm.weight", f"{mx}.mlp_ln.weight")
        put(f"{hf}.final_layer_norm.bias", f"{mx}.mlp_ln.bias")

    for i in range(dims.n_audio_layer):
        block(f"model.encoder.layers.{i}", f"encoder.blocks.{i}", False)
    put("model.encoder.layer_norm.weight", "encoder.ln_post.weight")
    put("model.encoder.layer_norm.bias", "encoder.ln_post.bias")
    put("model.decoder.embed_tokens.weight", "decoder.token_embedding.weight")
    put("model.decoder.embed_positions.weight", "decoder.positional_embedding")
    for i in range(dims.n_text_layer):
        block(f"model.decoder.layers.{i}", f"decoder.blocks.{i}", True)
    put("model.decoder.layer_norm.weight", "decoder.ln.weight")
    put("model.decoder.layer_norm.bias", "decoder.ln.bias")
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    missing = [k for k in missing if "k_proj.bias" not in k]
    assert not missing and not unexpected, (missing, unexpected)
    return m


def golden_mel():
    from transformers.models.whisper.feature_extraction_whisper import WhisperFeatureExtractor

    out = {}
    for n_mels in (80, 128):
        fe = WhisperFeatureExtractor(feature_size=n_mels)
        out[f"filters_{n_mels}"] = fe.mel_filters.T.astype(np.float32)  # [n_mels, 201]
        for name, idx, secs in (("full", 0, 30.0), ("short", 1, 5.0)):
            audio = R.synthetic_clip(idx, secs)
            mel = fe._np_extract_fbank_features(audio[None].astype(np.float64), "cpu")[0]  # [n_mels, 3000]
            mel = mel.T.astype(np.float32)  # time-major like the reference
            rows = np.r_[0:8, 1496:1504, 2992:3000]
            out[f"{name}_{n_mels}_rows"] = rows
            out[f"{name}_{n_mels}_slices"] = mel[rows]
            out[f"{name}_{n_mels}_stats"] = np.array(
                [mel.mean(), mel.std(), np.abs(mel).sum(), mel.max(), mel.min()], dtype=np.float64
            )
            out[f"{name}_{n_mels}_colmean"] = mel.mean(axis=0)
    np.savez_compressed(os.path.join(OUT, "mel.npz"), **out)
    print("mel.npz written")


def golden_model():
    dims = MICRO
    W = R.synthetic_weights(dims, seed=7)
    m = hf_model(dims, W)
    sp = R.SpecialTokens.multilingual()
    mels = np.stack([R.log_mel_spectrogram(R.synthetic_clip(0, 30.0)), R.log_mel_spectrogram(R.synthetic_clip(1, 5.0))])
    mel_t = torch.from_numpy(mels)
    out = {"mel_checksum": np.array([float(np.abs(mels).sum())])}
    with torch.no_grad():
        enc = m.model.encoder(mel_t.transpose(1, 2)).last_hidden_state  # [2,1500,d]
        out["enc_rows"] = np.r_[0:4, 748:752, 1496:1500]
        out["enc_slices"] = enc[:, out["enc_rows"]].numpy()
        out["enc_stats"] = np.array([enc.mean().item(), enc.std().item(), enc.abs().sum().item()])
        # teacher-forced logits on a framed token batch (EOT padded, ipa_data_loader.py:102-131)
        rng = np.random.default_rng(5)
        body = [rng.integers(0, 50257, size=n).tolist() for n in (9, 5)]
        sot = list(sp.sot_sequence_including_notimestamps(0))
        seqs = [sot + b + [sp.eot] for b in body]
        L = max(len(s) for s in seqs)
        tokens = np.array([s + [sp.eot] * (L - len(s)) for s in seqs], dtype=np.int64)
        out["tokens"] = tokens
        tt = torch.from_numpy(tokens)
        logits = m(encoder_outputs=(enc,), decoder_input_ids=tt[:, :-1]).logits  # [2, L-1, V]
        cols = np.r_[0:48, 220:224, 50250:50270, 50355:50370]
        out["logit_cols"] = cols
        out["logit_slices"] = logits[:, :, cols].numpy()
        out["logit_stats"] = np.array([logits.mean().item(), logits.std().item(), logits.abs().max().item()])
        tgt = tt[:, 1:]
        mask = R.loss_mask(tgt, sp.eot)
        ce = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), tgt.reshape(-1), reduction="none")
        out["loss"] = np.array([(ce * mask.reshape(-1)).sum().item() / max(int(mask.sum()), 1)])
        out["loss_mask"] = mask.numpy()
        # greedy decode WITHOUT a kv cache (full prefix re-run each step): checks the oracle's cache logic
        always, first = R.suppress_lists(sp)
        toks = torch.tensor([sot, sot], dtype=torch.long)
        margins = []
        nstep = 12
        for i in range(nstep):
            lg = m(encoder_outputs=(enc,), decoder_input_ids=toks).logits[:, -1].clone()
            if i == 0:
                lg[:, first] = float("-inf")
            lg[:, always] = float("-inf")
            t2 = torch.topk(lg, 2, dim=-1).values
            margins.append((t2[:, 0] - t2[:, 1]).numpy())
            nxt = lg.argmax(-1)
            prev_eot = toks[:, -1] == sp.eot
            nxt = torch.where(prev_eot, torch.full_like(nxt, sp.eot), nxt)
            toks = torch.cat([toks, nxt[:, None]], dim=1)
        out["greedy_tokens"] = toks.numpy()
        out["greedy_margins"] = np.stack(margins, axis=1)
        # language detection known answer
        lg = m(encoder_outputs=(enc,), decoder_input_ids=torch.full((2, 1), sp.sot)).logits[:, 0]
        lang = lg[:, sp.lang_first : sp.lang_first + sp.n_langs].argmax(-1) + sp.lang_first
        out["lang_tokens"] = lang.numpy()
    np.savez_compressed(os.path.join(OUT, "micro_model.npz"), **out)
    print("micro_model.npz written; greedy:", out["greedy_tokens"][:, 4:].tolist(), "min margin", out["greedy_margins"].min())


# ---------------------------------------------------------------------------------------------------------------------
# round 3: fixtures beyond MICRO dims -- whisper-small width, whisper-large-v3 dims (128 mels,
# 51 866 tokens, 100 languages), a masked-CE GRADIENT fixture from the stand-in's autograd, the fp16-features decode path
#  and the fp16-rounded sinusoid table (App. C.3).  One encoder + one decoder layer each, so the stand-in
# runs in seconds and the fixtures stay small; seeds / clips are the oracle's generators.
SMALL1 = R.ModelDimensions(80, 1500, 768, 12, 1, 51865, 448, 768, 12, 1)
LARGE1 = R.ModelDimensions(128, 1500, 1280, 20, 1, 51866, 448, 1280, 20, 1)
WIDE = {"small1": (SMALL1, 41, 99), "large1": (LARGE1, 43, 100)}  # name -> (dims, weight seed, languages)


def framed_tokens(sp, lens, seed):
    rng = np.random.default_rng(seed)
    sot = list(sp.sot_sequence_including_notimestamps(0))
    seqs = [sot + rng.integers(0, 50257, size=n).tolist() + [sp.eot] for n in lens]
    L = max(len(s) for s in seqs)
    return np.array([s + [sp.eot] * (L - len(s)) for s in seqs], dtype=np.int64)


GRAD_SLICES = {  # oracle name -> (HF name, row slice, column slice)
    "decoder.token_embedding.weight": ("model.decoder.embed_tokens.weight", "tok_rows", slice(0, 16)),
    "decoder.positional_embedding": ("model.decoder.embed_positions.weight", slice(0, 12), slice(0, 16)),
    "decoder.blocks.0.attn.query.weight": ("model.decoder.layers.0.self_attn.q_proj.weight", slice(0, 8), slice(0, 16)),
    "decoder.blocks.0.cross_attn.key.weight": ("model.decoder.layers.0.encoder_attn.k_proj.weight", slice(0, 8), slice(0, 16)),
    "decoder.blocks.0.cross_attn.value.bias": ("model.decoder.layers.0.encoder_attn.v_proj.bias", slice(0, 32), None),
    "decoder.blocks.0.mlp1.weight": ("model.decoder.layers.0.fc1.weight", slice(0, 8), slice(0, 16)),
    "decoder.blocks.0.mlp_ln.weight": ("model.decoder.layers.0.final_layer_norm.weight", slice(0, 32), None),
    "decoder.ln.bias": ("model.decoder.layer_norm.bias", slice(0, 32), None),
}


def golden_wide():
    out = {}
    for name, (dims, seed, n_lang) in WIDE.items():
        W = R.synthetic_weights(dims, seed=seed)
        m = hf_model(dims, W)
        sp = R.SpecialTokens.multilingual(n_lang)
        clips = [R.synthetic_clip(2, 30.0), R.synthetic_clip(3, 7.0)]
        mels = np.stack([R.log_mel_spectrogram(a, dims.n_mels) for a in clips])
        out[f"{name}_mel_checksum"] = np.array([float(np.abs(mels).sum())])
        tokens = framed_tokens(sp, (11, 6), seed)
        out[f"{name}_tokens"] = tokens
        tt = torch.from_numpy(tokens)
        with torch.no_grad():
            enc = m.model.encoder(torch.from_numpy(mels).transpose(1, 2)).last_hidden_state
        rows = np.r_[0:3, 749:752, 1497:1500]
        ecols = np.r_[0:24, dims.n_audio_state - 24:dims.n_audio_state]
        out[f"{name}_enc_rows"], out[f"{name}_enc_cols"] = rows, ecols
        out[f"{name}_enc_slices"] = enc[:, rows][:, :, ecols].numpy()
        out[f"{name}_enc_stats"] = np.array([enc.mean().item(), enc.std().item(), enc.abs().sum().item()])
        # teacher-forced logits, masked CE and its GRADIENT w.r.t. decoder parameters (stand-in autograd)
        for p_ in m.parameters():
            p_.requires_grad_(False)
        leaves = {}
        sd = dict(m.named_parameters())
        for oname, (hname, _, _) in GRAD_SLICES.items():
            sd[hname].requires_grad_(True)
            leaves[oname] = sd[hname]
        logits = m(encoder_outputs=(enc,), decoder_input_ids=tt[:, :-1]).logits
        cols = np.r_[0:32, 220:224, 50250:50270, 50355:50370, dims.n_vocab - 8:dims.n_vocab]
        out[f"{name}_logit_cols"] = cols
        out[f"{name}_logit_slices"] = logits[:, :, cols].detach().numpy()
        out[f"{name}_logit_stats"] = np.array([logits.mean().item(), logits.std().item(), logits.abs().max().item()])
        tgt = tt[:, 1:]
        mask = R.loss_mask(tgt, sp.eot)
        ce = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), tgt.reshape(-1), reduction="none")
        loss = (ce * mask.reshape(-1)).sum() / max(int(mask.sum()), 1)
        out[f"{name}_loss"] = np.array([loss.item()])
        grads = torch.autograd.grad(loss, list(leaves.values()))
        tok_rows = np.unique(tokens[:, :-1])[:12]
        out[f"{name}_grad_tok_rows"] = tok_rows
        for (oname, (hname, rs, cs)), g in zip(GRAD_SLICES.items(), grads):
            if isinstance(rs, str):
                rs = tok_rows
            sl = g[rs] if cs is None else g[rs][:, cs]
            key = oname.replace(".", "__")
            out[f"{name}_grad__{key}"] = sl.numpy()
            out[f"{name}_gradnorm__{key}"] = np.array([float(g.norm())])
        with torch.no_grad():
            # language detection known answer at this vocabulary (99 / 100 languages)
            lg = m(encoder_outputs=(enc,), decoder_input_ids=torch.full((2, 1), sp.sot)).logits[:, 0]
            out[f"{name}_lang_tokens"] = (lg[:, sp.lang_first: sp.lang_first + sp.n_langs].argmax(-1) + sp.lang_first).numpy()
            # greedy ids without a KV cache, on f32 features and on fp16-ROUNDED features (DecodingOptions.fp16=True of
            # transcribe_single.py:49-52: the encoder output is cast to fp16 before cross-attention)
            always, first = R.suppress_lists(sp)
            sot = list(sp.sot_sequence_including_notimestamps(0))
            for tag, feats in (("f32", enc), ("fp16feat", enc.half().float())):
                toks = torch.tensor([sot, sot], dtype=torch.long)
                margins = []
                for i in range(8):
                    l2 = m(encoder_outputs=(feats,), decoder_input_ids=toks).logits[:, -1].clone()
                    if i == 0:
                        l2[:, first] = float("-inf")
                    l2[:, always] = float("-inf")
                    t2 = torch.topk(l2, 2, dim=-1).values
                    margins.append((t2[:, 0] - t2[:, 1]).numpy())
                    toks = torch.cat([toks, l2.argmax(-1)[:, None]], dim=1)
                out[f"{name}_greedy_{tag}_tokens"] = toks.numpy()
                out[f"{name}_greedy_{tag}_margins"] = np.stack(margins, axis=1)
                if tag == "fp16feat":
                    out[f"{name}_fp16feat_last_logit_slices"] = l2[:, cols[:32]].numpy()
            # mlx_whisper builds the encoder's sinusoid table in the load dtype (fp16) and set_dtype(float32)
            # does not touch it [UPSTREAM-UNVERIFIED]: the encoder output with the table rounded to fp16
            m.model.encoder.embed_positions.weight.copy_(R.sinusoids(dims.n_audio_ctx, dims.n_audio_state).half().float())
            enc16 = m.model.encoder(torch.from_numpy(mels).transpose(1, 2)).last_hidden_state
            out[f"{name}_enc_fp16pos_slices"] = enc16[:, rows][:, :, ecols].numpy()
            out[f"{name}_enc_fp16pos_maxdiff"] = np.array([(enc16 - enc).abs().max().item()])
        print(name, "done: loss", float(loss), "greedy", out[f"{name}_greedy_f32_tokens"][:, 4:].tolist(),
              "fp16-feature ids equal:", bool((out[f"{name}_greedy_f32_tokens"] == out[f"{name}_greedy_fp16feat_tokens"]).all()),
              "fp16 sinusoids move the features by", float(out[f"{name}_enc_fp16pos_maxdiff"][0]))
    path = os.path.join(OUT, "wide_model.npz")
    np.savez_compressed(path, **{k: (v.astype(np.float32) if v.dtype == np.float64 and v.size > 8 else v) for k, v in out.items()})
    print("wide_model.npz written,", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["mel", "micro", "wide"]
    if "mel" in which:
        golden_mel()
    if "micro" in which:
        golden_model()
    if "wide" in which:
        golden_wide()
