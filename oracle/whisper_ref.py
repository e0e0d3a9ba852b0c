"""CPU oracle for the Whisper -> IPA hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product path (``whisper_ipa_amd``) never imports it and raises when
``libwipa.so`` is missing.

What it restates
----------------
The reference (barathanaslan/whisper-ipa) owns none of the arithmetic on this
path: it calls ``mlx_whisper==0.4.3`` / ``mlx==0.30.0`` (requirements.txt:21-23),
which are NOT in /root/reference and not installable here.  So this is a plain
torch-CPU fp32 restatement of the *published* mlx_whisper algorithm (which
mirrors openai/whisper), anchored on the reference's own call sites:

* log-mel front-end ........ scripts/ipa_data_loader.py:79-85,
                             scripts/transcribe_single.py:43-47
* encoder / embed_audio .... scripts/train_whisper_ipa.py:223,
                             scripts/transcribe_single.py:54
* teacher-forced logits .... scripts/train_whisper_ipa.py:228-232
* masked CE loss ........... scripts/train_whisper_ipa.py:207-263
* clip_grad_dict + AdamW ... scripts/train_whisper_ipa.py:287-306,513
                             (the clip walks dicts only: decoder.blocks, a list,
                             passes through unclipped -- clip_grad_dict below)
* greedy decode ............ scripts/transcribe_single.py:49-56,
                             scripts/train_whisper_ipa.py:338-356
* token framing / EOT pad .. scripts/ipa_data_loader.py:102-131

PARITY PIN STATUS: **parity unpinned against the true reference** -- the
reference holds no golden mel / logits / loss / token-id vector for this path
 and mlx cannot run here.  The oracle is pinned instead
against a stand-in of independent lineage: the locally installed
``transformers`` Whisper classes, on seeded synthetic weights/audio, via the
fixtures in tests/golden/ (generator: tools/make_golden.py).

All tensors are torch CPU float32 unless noted.  Weight names/layouts follow
the mlx_whisper checkpoint contract the reference writes
(train_whisper_ipa.py:43-57,421): Linear ``[out, in]``, Conv1d
``[C_out, K, C_in]`` (channels-last), flat dotted keys.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
CHUNK_LENGTH = 30
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE  # 480000
N_FRAMES = N_SAMPLES // HOP_LENGTH  # 3000


@dataclass
class ModelDimensions:
    n_mels: int
    n_audio_ctx: int
    n_audio_state: int
    n_audio_head: int
    n_audio_layer: int
    n_vocab: int
    n_text_ctx: int
    n_text_state: int
    n_text_head: int
    n_text_layer: int


DIMS = {
    "tiny": ModelDimensions(80, 1500, 384, 6, 4, 51865, 448, 384, 6, 4),
    "base": ModelDimensions(80, 1500, 512, 8, 6, 51865, 448, 512, 8, 6),
    "small": ModelDimensions(80, 1500, 768, 12, 12, 51865, 448, 768, 12, 12),
    "medium": ModelDimensions(80, 1500, 1024, 16, 24, 51865, 448, 1024, 16, 24),
    "large-v3": ModelDimensions(128, 1500, 1280, 20, 32, 51866, 448, 1280, 20, 32),
}

# ---------------------------------------------------------------------------
# audio front-end  (mlx_whisper.audio; call sites ipa_data_loader.py:80-82)
# ---------------------------------------------------------------------------


def pad_or_trim(audio: np.ndarray, length: int = N_SAMPLES) -> np.ndarray:
    """Zero-pad or cut the last axis to ``length`` (ipa_data_loader.py:80)."""
    audio = np.asarray(audio, dtype=np.float32)
    n = audio.shape[-1]
    if n > length:
        return audio[..., :length]
    if n < length:
        pad = [(0, 0)] * (audio.ndim - 1) + [(0, length - n)]
        return np.pad(audio, pad)
    return audio


def _hz_to_mel_slaney(f: np.ndarray) -> np.ndarray:
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    with np.errstate(divide="ignore"):
        log_t = min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep
    return np.where(f >= min_log_hz, log_t, mels)


def _mel_to_hz_slaney(m: np.ndarray) -> np.ndarray:
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filters(n_mels: int) -> np.ndarray:
    """librosa.filters.mel(sr=16000, n_fft=400, n_mels, fmin=0, fmax=8000,
    htk=False, norm='slaney') -- the table mlx_whisper ships as
    assets/mel_filters.npz.  Returns [n_mels, 201] float32."""
    n_freqs = N_FFT // 2 + 1
    fft_freqs = np.linspace(0.0, SAMPLE_RATE / 2, n_freqs)
    mel_pts = np.linspace(_hz_to_mel_slaney(0.0), _hz_to_mel_slaney(8000.0), n_mels + 2)
    hz_pts = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(hz_pts)
    ramps = hz_pts[:, None] - fft_freqs[None, :]
    w = np.zeros((n_mels, n_freqs), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (hz_pts[2 : n_mels + 2] - hz_pts[:n_mels])
    w *= enorm[:, None]
    return w.astype(np.float32)


def hann_periodic(n: int = N_FFT) -> np.ndarray:
    return np.hanning(n + 1)[:-1].astype(np.float32)


def log_mel_spectrogram(audio: np.ndarray, n_mels: int = 80) -> np.ndarray:
    """[n] f32 -> [n_frames, n_mels] f32, time-major like mlx_whisper
    (ipa_data_loader.py:82-85: "returns (n_frames, n_mels) = (3000, n_mels)").

    reflect-pad 200, periodic Hann(400), hop 160, rFFT, |.|^2 of all frames but
    the last, mel filterbank, log10(max(.,1e-10)), max(., global_max - 8),
    (. + 4) / 4.  Everything in fp32, as the reference's MLX path.
    """
    x = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32))
    pad = N_FFT // 2
    xp = F.pad(x[None, None, :], (pad, pad), mode="reflect")[0, 0]
    frames = xp.unfold(0, N_FFT, HOP_LENGTH)  # [n_frames+1, 400]
    win = torch.from_numpy(hann_periodic())
    spec = torch.fft.rfft(frames * win, dim=-1)  # complex64
    mag = (spec.real**2 + spec.imag**2)[:-1]  # drop last frame
    filt = torch.from_numpy(mel_filters(n_mels))
    mel = mag @ filt.T
    log_spec = torch.log10(torch.clamp(mel, min=1e-10))
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    log_spec = (log_spec + 4.0) / 4.0
    return log_spec.numpy()


# ---------------------------------------------------------------------------
# model  (mlx_whisper.whisper)
# ---------------------------------------------------------------------------


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    assert channels % 2 == 0
    log_inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-log_inc * torch.arange(channels // 2, dtype=torch.float32))
    t = torch.arange(length, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([torch.sin(t), torch.cos(t)], dim=1)


def _linear(x: torch.Tensor, W: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    b = W.get(prefix + ".bias")
    return F.linear(x, W[prefix + ".weight"], b)


def _layer_norm(x: torch.Tensor, W: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), W[prefix + ".weight"], W[prefix + ".bias"], 1e-5)


def _mha(
    x: torch.Tensor,
    W: Dict[str, torch.Tensor],
    prefix: str,
    n_head: int,
    xa: Optional[torch.Tensor] = None,
    mask: Optional[torch.Tensor] = None,
    kv: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
) -> Tuple[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
    """MultiHeadAttention of mlx_whisper: key has no bias; q and k are EACH
    scaled by head_dim**-0.25; softmax in fp32; self-attn cache concatenates on
    the time axis, cross-attn computes k,v from xa once and then reuses them."""
    q = _linear(x, W, prefix + ".query")
    if xa is None:
        k = _linear(x, W, prefix + ".key")
        v = _linear(x, W, prefix + ".value")
        if kv is not None:
            k = torch.cat([kv[0], k], dim=1)
            v = torch.cat([kv[1], v], dim=1)
    elif kv is None:
        k = _linear(xa, W, prefix + ".key")
        v = _linear(xa, W, prefix + ".value")
    else:
        k, v = kv
    B, Tq, D = q.shape
    Tk = k.shape[1]
    hd = D // n_head
    scale = hd**-0.25
    qh = (q.view(B, Tq, n_head, hd).permute(0, 2, 1, 3)) * scale
    kh = (k.view(B, Tk, n_head, hd).permute(0, 2, 3, 1)) * scale
    vh = v.view(B, Tk, n_head, hd).permute(0, 2, 1, 3)
    qk = qh @ kh
    if mask is not None:
        qk = qk + mask[Tk - Tq : Tk, :Tk]
    w = torch.softmax(qk.float(), dim=-1)
    out = (w @ vh).permute(0, 2, 1, 3).reshape(B, Tq, D)
    return _linear(out, W, prefix + ".out"), (k, v)


def _gelu(x: torch.Tensor) -> torch.Tensor:
    return F.gelu(x)  # exact erf form (mlx nn.gelu)


def _block(
    x: torch.Tensor,
    W: Dict[str, torch.Tensor],
    prefix: str,
    n_head: int,
    xa: Optional[torch.Tensor] = None,
    mask: Optional[torch.Tensor] = None,
    cache: Optional[dict] = None,
) -> torch.Tensor:
    kv_self = cache.get("self") if cache is not None else None
    y, kv_self = _mha(_layer_norm(x, W, prefix + ".attn_ln"), W, prefix + ".attn", n_head, mask=mask, kv=kv_self)
    x = x + y
    if cache is not None:
        cache["self"] = kv_self
    if xa is not None:
        kv_cross = cache.get("cross") if cache is not None else None
        y, kv_cross = _mha(
            _layer_norm(x, W, prefix + ".cross_attn_ln"), W, prefix + ".cross_attn", n_head, xa=xa, kv=kv_cross
        )
        x = x + y
        if cache is not None:
            cache["cross"] = kv_cross
    h = _gelu(_linear(_layer_norm(x, W, prefix + ".mlp_ln"), W, prefix + ".mlp1"))
    return x + _linear(h, W, prefix + ".mlp2")


def encoder_forward(
    W: Dict[str, torch.Tensor], dims: ModelDimensions, mel: torch.Tensor, n_layers: Optional[int] = None
) -> torch.Tensor:
    """AudioEncoder.__call__: mel [B, 3000, n_mels] -> [B, 1500, d].
    (train_whisper_ipa.py:223 ``model.embed_audio``; transcribe_single.py:54)."""
    x = mel.transpose(1, 2)  # [B, C, L]
    w1 = W["encoder.conv1.weight"].permute(0, 2, 1)  # [C_out,K,C_in] -> [C_out,C_in,K]
    w2 = W["encoder.conv2.weight"].permute(0, 2, 1)
    x = _gelu(F.conv1d(x, w1, W["encoder.conv1.bias"], padding=1))
    x = _gelu(F.conv1d(x, w2, W["encoder.conv2.bias"], stride=2, padding=1))
    x = x.transpose(1, 2)  # [B, 1500, d]
    pos = W.get("encoder._positional_embedding")
    if pos is None:
        pos = sinusoids(dims.n_audio_ctx, dims.n_audio_state)
    x = x + pos[: x.shape[1]]
    L = dims.n_audio_layer if n_layers is None else n_layers
    for i in range(L):
        x = _block(x, W, f"encoder.blocks.{i}", dims.n_audio_head)
    return _layer_norm(x, W, "encoder.ln_post")


def causal_mask(n: int) -> torch.Tensor:
    return torch.triu(torch.full((n, n), float("-inf")), diagonal=1)


def decoder_forward(
    W: Dict[str, torch.Tensor],
    dims: ModelDimensions,
    tokens: torch.Tensor,
    xa: torch.Tensor,
    cache: Optional[List[dict]] = None,
) -> torch.Tensor:
    """TextDecoder.__call__: tokens [B, T] int, xa [B, 1500, d] -> logits [B, T, V].
    With ``cache`` (list of per-layer dicts) the positional offset is the cached
    self-attention length and the new K/V are appended (KV-cached decode);
    without it this is the teacher-forced ``model.logits`` of
    train_whisper_ipa.py:232."""
    offset = 0
    if cache is not None and "self" in cache[0]:
        offset = cache[0]["self"][0].shape[1]
    T = tokens.shape[1]
    x = W["decoder.token_embedding.weight"][tokens] + W["decoder.positional_embedding"][offset : offset + T]
    mask = causal_mask(max(dims.n_text_ctx, offset + T))
    for i in range(dims.n_text_layer):
        c = cache[i] if cache is not None else None
        x = _block(x, W, f"decoder.blocks.{i}", dims.n_text_head, xa=xa, mask=mask, cache=c)
    x = _layer_norm(x, W, "decoder.ln")
    return x @ W["decoder.token_embedding.weight"].T


# ---------------------------------------------------------------------------
# tokenizer constants (mlx_whisper.tokenizer, multilingual <= large-v2 numbering;
# WHISPER_IPA_RESEARCH_STANDALONE.md:333-338)
# ---------------------------------------------------------------------------


@dataclass
class SpecialTokens:
    eot: int
    sot: int
    lang_first: int
    n_langs: int
    translate: int
    transcribe: int
    sot_lm: int
    sot_prev: int
    no_speech: int
    no_timestamps: int
    timestamp_begin: int

    @staticmethod
    def multilingual(num_languages: int = 99) -> "SpecialTokens":
        eot = 50257
        sot = 50258
        lang_first = 50259
        translate = lang_first + num_languages
        return SpecialTokens(
            eot, sot, lang_first, num_languages, translate, translate + 1, translate + 2, translate + 3,
            translate + 4, translate + 5, translate + 6,
        )

    def sot_sequence_including_notimestamps(self, lang_index: int = 0) -> Tuple[int, ...]:
        return (self.sot, self.lang_first + lang_index, self.transcribe, self.no_timestamps)


# Ids of tokenizer.non_speech_tokens for the multilingual (<= large-v2) vocabulary
# (published list; same numbers as transformers' NON_SPEECH_TOKENS_MULTI minus the
# task tokens the decoder appends itself).
NON_SPEECH_TOKENS_MULTI = [
    1, 2, 7, 8, 9, 10, 14, 25, 26, 27, 28, 29, 31, 58, 59, 60, 61, 62, 63, 90, 91, 92, 93, 359, 503, 522, 542, 873,
    893, 902, 918, 922, 931, 1350, 1853, 1982, 2460, 2627, 3246, 3253, 3268, 3536, 3846, 3961, 4183, 4667, 6585, 6647,
    7273, 9061, 9383, 10428, 10929, 11938, 12033, 12331, 12562, 13793, 14157, 14635, 15265, 15618, 16553, 16604, 18362,
    18956, 20075, 21675, 22520, 26130, 26161, 26435, 28279, 29464, 31650, 32302, 32470, 36865, 42863, 47425, 49870,
    50254,
]
BLANK_TOKEN = 220  # tokenizer.encode(" ")


def suppress_lists(sp: SpecialTokens, non_speech: Sequence[int] = NON_SPEECH_TOKENS_MULTI) -> Tuple[List[int], List[int]]:
    """(suppressed at every step, additionally suppressed at the first step)
    for DecodingOptions(suppress_tokens="-1", suppress_blank=True,
    without_timestamps=True) -- the options of transcribe_single.py:49-52."""
    always = sorted(set(list(non_speech) + [sp.transcribe, sp.translate, sp.sot, sp.sot_prev, sp.sot_lm, sp.no_speech]))
    first = [BLANK_TOKEN, sp.eot]
    return always, first


@dataclass
class GreedyResult:
    tokens: np.ndarray  # [B, n_init + n_steps] int64, EOT-latched
    n_steps: int
    sum_logprobs: np.ndarray  # [B]
    margins: np.ndarray  # [B, n_steps] top1 - top2 of the filtered logits
    step_logits: Optional[np.ndarray] = None  # [B, n_steps, V] when requested


def greedy_decode(
    W: Dict[str, torch.Tensor],
    dims: ModelDimensions,
    audio_features: torch.Tensor,
    initial_tokens: Sequence[int],
    suppress_always: Sequence[int],
    suppress_first: Sequence[int],
    eot: int,
    sample_len: Optional[int] = None,
    stop_on_eot: bool = True,
    keep_logits: bool = False,
    fp16_features: bool = False,
) -> GreedyResult:
    """DecodingTask._main_loop with GreedyDecoder(temperature=0), n_group=1
    (transcribe_single.py:55; train_whisper_ipa.py:356).

    step 0 feeds all initial tokens, later steps only the last one (KV cache);
    logits of the last position; SuppressBlank at step 0; SuppressTokens at
    every step; argmax; rows whose previous token was EOT stay EOT; stop when
    every row ended or the context is full.  ``fp16_features`` reproduces
    DecodingOptions.fp16=True of transcribe_single.py (features rounded to
    fp16 before cross-attention).
    """
    B = audio_features.shape[0]
    if sample_len is None:
        sample_len = dims.n_text_ctx // 2
    xa = audio_features.half().float() if fp16_features else audio_features
    tokens = torch.tensor([list(initial_tokens)] * B, dtype=torch.long)
    cache: List[dict] = [dict() for _ in range(dims.n_text_layer)]
    always = torch.tensor(list(suppress_always), dtype=torch.long)
    first = torch.tensor(list(suppress_first), dtype=torch.long)
    sum_lp = torch.zeros(B)
    margins, kept = [], []
    n_steps = 0
    for i in range(sample_len):
        inp = tokens if i == 0 else tokens[:, -1:]
        logits = decoder_forward(W, dims, inp, xa, cache)[:, -1].float()
        if i == 0 and len(first):
            logits[:, first] = float("-inf")
        if len(always):
            logits[:, always] = float("-inf")
        if keep_logits:
            kept.append(logits.clone())
        top2 = torch.topk(logits, 2, dim=-1).values
        margins.append((top2[:, 0] - top2[:, 1]))
        nxt = logits.argmax(dim=-1)
        logprobs = torch.log_softmax(logits, dim=-1)
        cur = logprobs[torch.arange(B), nxt]
        prev_eot = tokens[:, -1] == eot
        sum_lp = sum_lp + cur * (~prev_eot)
        nxt = torch.where(prev_eot, torch.full_like(nxt, eot), nxt)
        tokens = torch.cat([tokens, nxt[:, None]], dim=1)
        n_steps += 1
        if (stop_on_eot and bool((tokens[:, -1] == eot).all())) or tokens.shape[1] > dims.n_text_ctx:
            break
    return GreedyResult(
        tokens.numpy(),
        n_steps,
        sum_lp.numpy(),
        torch.stack(margins, dim=1).numpy(),
        torch.stack(kept, dim=1).numpy() if keep_logits else None,
    )


def detect_language(
    W: Dict[str, torch.Tensor], dims: ModelDimensions, audio_features: torch.Tensor, sp: SpecialTokens
) -> np.ndarray:
    """Whisper.detect_language (train_whisper_ipa.py:339, language=None): one
    decoder pass on [sot], everything but the language tokens masked, argmax."""
    B = audio_features.shape[0]
    tokens = torch.full((B, 1), sp.sot, dtype=torch.long)
    logits = decoder_forward(W, dims, tokens, audio_features)[:, 0].float()
    mask = torch.ones(logits.shape[-1], dtype=torch.bool)
    mask[sp.lang_first : sp.lang_first + sp.n_langs] = False
    logits[:, mask] = float("-inf")
    return logits.argmax(dim=-1).numpy()


# ---------------------------------------------------------------------------
# loss / clip / optimiser  (train_whisper_ipa.py:207-311)
# ---------------------------------------------------------------------------


def loss_mask(target: torch.Tensor, eot: int) -> torch.Tensor:
    """mask = (tgt != eot) | (cumsum(tgt == eot) == 1)   (train_whisper_ipa.py:242-247)."""
    is_eot = target == eot
    return (~is_eot) | (torch.cumsum(is_eot.long(), dim=1) == 1)


def compute_loss(
    W: Dict[str, torch.Tensor], dims: ModelDimensions, mel: torch.Tensor, tokens: torch.Tensor, eot: int
) -> torch.Tensor:
    """train_whisper_ipa.py:207-263: encoder (frozen), teacher-forced decoder on
    tokens[:, :-1], CE(reduction='none') against tokens[:, 1:], masked mean with
    the batch-global valid count."""
    with torch.no_grad():
        xa = encoder_forward(W, dims, mel)
    return loss_from_features(W, dims, xa, tokens, eot)


def loss_from_features(
    W: Dict[str, torch.Tensor], dims: ModelDimensions, xa: torch.Tensor, tokens: torch.Tensor, eot: int
) -> torch.Tensor:
    dec_in, tgt = tokens[:, :-1], tokens[:, 1:]
    logits = decoder_forward(W, dims, dec_in, xa)
    mask = loss_mask(tgt, eot).reshape(-1)
    ce = F.cross_entropy(logits.reshape(-1, logits.shape[-1]).float(), tgt.reshape(-1), reduction="none")
    masked = torch.where(mask, ce, torch.zeros_like(ce))
    return masked.sum() / torch.clamp(mask.sum(), min=1)


def clip_per_tensor(g: torch.Tensor, max_norm: float = 1.0) -> torch.Tensor:
    """train_whisper_ipa.py:295-298 -- the arithmetic applied to ONE gradient array: L2 norm, coefficient
    max_norm / (norm + 1e-6) capped at 1, scale."""
    norm = torch.sqrt(torch.sum(g * g))
    coef = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
    return g * coef


def unflatten_params(flat: Dict[str, torch.Tensor]):
    """Flat dotted keys -> the NESTED tree mlx hands to ``clip_grad_dict``: a dict per module, and a Python LIST wherever
    every key at a level is an integer (``decoder.blocks`` is a list of blocks in mlx_whisper; the reference's own
    ``flatten_params`` has the matching ``isinstance(params, list)`` branch, train_whisper_ipa.py:43-57, which is where the
    ``decoder.blocks.{i}.`` checkpoint keys come from)."""
    root: dict = {}
    for key, val in flat.items():
        node = root
        parts = key.split(".")
        for part in parts[:-1]:
            node = node.setdefault(part, {})
        node[parts[-1]] = val

    def listify(node):
        if not isinstance(node, dict):
            return node
        node = {k: listify(v) for k, v in node.items()}
        if node and all(k.isdigit() for k in node):
            return [node[str(i)] for i in range(len(node))]
        return node

    return listify(root)


def flatten_params(params, prefix: str = "") -> Dict[str, torch.Tensor]:
    """train_whisper_ipa.py:43-57: dict -> dotted keys, list -> index keys."""
    flat: Dict[str, torch.Tensor] = {}
    if isinstance(params, dict):
        for k, v in params.items():
            flat.update(flatten_params(v, f"{prefix}.{k}" if prefix else k))
    elif isinstance(params, list):
        for i, v in enumerate(params):
            flat.update(flatten_params(v, f"{prefix}.{i}" if prefix else str(i)))
    elif prefix:
        flat[prefix] = params
    return flat


def clip_grad_dict(grad_dict: dict, max_norm: float = 1.0) -> dict:
    """train_whisper_ipa.py:287-303 AS WRITTEN: walk the ``dict`` values only -- a dict recurses (:290-291), anything with a
    ``.shape`` is clipped by its own L2 norm (:292-298), ANYTHING ELSE IS PASSED THROUGH (:299-300).  A ``list`` is neither a
    dict nor an array, so the whole ``decoder.blocks`` list -- every tensor of every decoder block -- leaves this function
    unclipped; what it clips is ``decoder.token_embedding.weight``, ``decoder.positional_embedding`` and
    ``decoder.ln.{weight,bias}``.  [UPSTREAM-UNVERIFIED only in that ``blocks`` is a list in mlx_whisper's TextDecoder; the
    reference's flatten_params and its checkpoint keys say so.]"""
    clipped = {}
    for key, value in grad_dict.items():
        if isinstance(value, dict):
            clipped[key] = clip_grad_dict(value, max_norm)
        elif hasattr(value, "shape"):
            clipped[key] = clip_per_tensor(value, max_norm)
        else:
            clipped[key] = value
    return clipped


def clip_gradients(grads: Dict[str, torch.Tensor], max_norm: float = 1.0, scope: str = "reference") -> Dict[str, torch.Tensor]:
    """Flat-keyed front of the two readings of the clip.  ``scope="reference"``: the tree walk of ``clip_grad_dict`` above
    (block tensors pass through).  ``scope="all"``: every tensor clipped by its own norm -- what the reference's docstring
    says it does, and what this oracle restated until round 5."""
    if scope == "reference":
        return flatten_params(clip_grad_dict(unflatten_params(grads), max_norm))
    if scope == "all":
        return {k: clip_per_tensor(g, max_norm) for k, g in grads.items()}
    raise ValueError(f"clip scope {scope!r}: 'reference' or 'all'")


def clipped_by_reference(name: str) -> bool:
    """True for the tensors ``clip_grad_dict`` reaches (no list on the path from the root)."""
    return not any(part.isdigit() for part in name.split("."))


def adamw_mlx(
    p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor,
    lr: float = 1e-5, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8, wd: float = 0.01,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """mlx.optimizers.AdamW defaults, no bias correction
    (train_whisper_ipa.py:513 ``optim.AdamW(learning_rate=lr)``)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    p = p * (1 - lr * wd) - lr * m / (torch.sqrt(v) + eps)
    return p, m, v


def train_step(
    W: Dict[str, torch.Tensor], dims: ModelDimensions, mel: torch.Tensor, tokens: torch.Tensor, eot: int,
    state: Dict[str, Tuple[torch.Tensor, torch.Tensor]], lr: float = 1e-5, max_grad_norm: float = 1.0,
    clip_scope: str = "reference",
) -> Tuple[float, Dict[str, torch.Tensor]]:
    """One reference training step on the decoder parameters (encoder frozen,
    train_whisper_ipa.py:181-204,266-311): loss + grads (:284), ``clip_grad_dict`` on the nested gradient tree (:287-303),
    AdamW (:306).  Updates ``W`` and ``state`` in place; returns (loss, grads after the clip)."""
    names = [k for k in W if k.startswith("decoder.")]
    leaves = {k: W[k].detach().clone().requires_grad_(True) for k in names}
    Wl = dict(W)
    Wl.update(leaves)
    loss = compute_loss(Wl, dims, mel, tokens, eot)
    grads = dict(zip(names, torch.autograd.grad(loss, [leaves[k] for k in names])))
    out = clip_gradients(grads, max_grad_norm, clip_scope)
    assert list(out) == names or set(out) == set(names)
    for k in names:
        g = out[k]
        m, v = state.get(k, (torch.zeros_like(g), torch.zeros_like(g)))
        p, m, v = adamw_mlx(W[k], g, m, v, lr=lr)
        W[k] = p.detach()
        state[k] = (m, v)
    return float(loss.detach()), out


# ---------------------------------------------------------------------------
# synthetic weights (shared by tests / bench; deterministic, no checkpoint needed)
# ---------------------------------------------------------------------------


PEAKY_GAIN = 128.0  # see peaky_positional_table


def peaky_positional_table(W: Dict[str, torch.Tensor], dims: ModelDimensions, seed: int, suppress_always: Sequence[int],
                           gain: float = PEAKY_GAIN) -> torch.Tensor:
    """The "peaky" preset's decoder.positional_embedding ("generate synthetic weights with a peaky
    output"): the lively table + gain * token_embedding[pi(p)], pi a seeded draw of distinct, never-suppressed text tokens.
    The residual stream carries that pointer through all blocks, the tied output embedding reads it back, and the token at
    position p + 1 is pi(p) with a top-1 margin of several logit standard deviations (whisper-small, gain 128: min margin
    14.6 over 2 x 64 steps against a logit std of 5.5 and a bf16 logit error of ~0.1-0.2) -- a CONFIDENT model, which is what a
    trained Whisper is and a random-init one is not.  What it decides the ids with is position, not audio: the audio (and the
    token history) still move every logit, which the logit-error tests measure, but they no longer pick the winner.  Used to
    show that the bf16 path reproduces the f32 ids bit for bit whenever margins exceed the arithmetic's error."""
    g = torch.Generator().manual_seed(seed + 12345)
    banned = set(int(t) for t in suppress_always)
    cand = torch.tensor([i for i in range(1000, 50000) if i not in banned])
    pi = cand[torch.randperm(len(cand), generator=g)[: dims.n_text_ctx]]
    return W["decoder.positional_embedding"] + gain * W["decoder.token_embedding.weight"][pi].float()


def synthetic_weights(
    dims: ModelDimensions, seed: int = 0, std: float = 0.06, emb_std: float = 0.2, pos_std: float = 1.2,
    out_scale: float = 4.0, preset: str = "lively",
) -> Dict[str, torch.Tensor]:
    """Seeded random-init Whisper weights in mlx_whisper naming.

    A plain std-0.02 init makes greedy decode degenerate (the tied embedding
    makes every step repeat the last prompt token), so the defaults are a
    "lively" preset: wide block weights, a wide positional table and the
    decoder out/mlp2 projections scaled with the embedding, which gives token
    sequences that vary with step and audio and logits of std ~2 (SURVEY.md
    section 7 hard parts).  Top-1 margins stay a fixed fraction of the logit
    spread, so low-precision parity tests gate on margin."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, s=std):
        return torch.randn(*shape, generator=g) * s

    W: Dict[str, torch.Tensor] = {}
    d = dims.n_audio_state
    W["encoder.conv1.weight"] = rn(d, 3, dims.n_mels, s=0.05)
    W["encoder.conv1.bias"] = rn(d)
    W["encoder.conv2.weight"] = rn(d, 3, d)
    W["encoder.conv2.bias"] = rn(d)

    def block(prefix, d, cross):
        names = ["attn"] + (["cross_attn"] if cross else [])
        for a in names:
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
    dt = dims.n_text_state
    W["decoder.token_embedding.weight"] = rn(dims.n_vocab, dt, s=emb_std)
    W["decoder.positional_embedding"] = rn(dims.n_text_ctx, dt, s=pos_std)
    for i in range(dims.n_text_layer):
        block(f"decoder.blocks.{i}", dt, True)
    for k in list(W):
        if k.startswith("decoder.blocks.") and k.split(".")[-2] in ("out", "mlp2"):
            W[k] = W[k] * out_scale
    W["decoder.ln.weight"] = 1.0 + rn(dt, s=0.1)
    W["decoder.ln.bias"] = rn(dt, s=0.1)
    if preset == "peaky":
        W["decoder.positional_embedding"] = peaky_positional_table(
            W, dims, seed, suppress_lists(SpecialTokens.multilingual(dims.n_vocab - 51765 - int(dims.n_vocab >= 51865)))[0])
    else:
        assert preset == "lively", preset
    return W


def synthetic_clip(idx: int, seconds: float = 30.0) -> np.ndarray:
    """BASELINE.md section 3 synthetic audio: default_rng(1234+idx), 0.1*N(0,1) f32;
    ``seconds`` < 30 gives the "short clip + zero padding" variant."""
    rng = np.random.default_rng(1234 + idx)
    n = int(round(seconds * SAMPLE_RATE))
    a = (0.1 * rng.standard_normal(n)).astype(np.float32)
    return pad_or_trim(a)
