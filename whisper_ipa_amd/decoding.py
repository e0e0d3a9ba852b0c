"""KV-cached greedy decoding with the surface of ``mlx_whisper.decoding`` used by the
reference: ``DecodingOptions``, ``DecodingResult``, ``decode(model, mel_or_features, options)``
(scripts/transcribe_single.py:49-56; scripts/train_whisper_ipa.py:338-362;
scripts/evaluate_model.py:170-201).

The whole loop runs on the GPU: one decoder step is captured into a hipGraph by
csrc/runtime.hip and replayed; token ids, EOT latches and log-prob sums stay in device
memory; the host only looks at the last token column every few steps to stop early.
"""
from __future__ import annotations

import os

import ctypes as C
import zlib
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from .runtime import on_stream, ptr, sptr, stream, stream_id
from .tokenizer import Tokenizer, get_tokenizer


@dataclass(frozen=True)
class DecodingOptions:
    task: str = "transcribe"
    language: Optional[str] = None
    temperature: float = 0.0
    sample_len: Optional[int] = None
    best_of: Optional[int] = None
    beam_size: Optional[int] = None
    patience: Optional[float] = None
    length_penalty: Optional[float] = None
    prompt: Optional[Union[str, List[int]]] = None
    prefix: Optional[Union[str, List[int]]] = None
    suppress_tokens: Optional[Union[str, Sequence[int]]] = "-1"
    suppress_blank: bool = True
    without_timestamps: bool = False
    max_initial_timestamp: Optional[float] = 1.0
    fp16: bool = True


@dataclass
class DecodingResult:
    audio_features: torch.Tensor
    language: str
    language_probs: Optional[dict] = None
    tokens: List[int] = field(default_factory=list)
    text: str = ""
    avg_logprob: float = float("nan")
    no_speech_prob: float = float("nan")
    temperature: float = float("nan")
    compression_ratio: float = float("nan")


@dataclass
class GreedyTokens:
    tokens: np.ndarray        # [B, n_init + n_steps] int64 (EOT-latched)
    n_steps: int              # generated positions
    sum_logprobs: np.ndarray  # [B]
    last_logits: Optional[torch.Tensor] = None  # [B, V] f32 logits of the last executed step (device)


class DecoderState:
    """The caller-owned decode blob of wipa_decoder_* (tokens, position, logits, KV caches)."""

    def __init__(self, model, B: int, pk=None):
        L = _lib.lib()
        self.model, self.B = model, B
        pk = pk or model.packed()
        self.layout = _lib.DecLayout()
        _lib.check(L.wipa_decoder_layout(C.byref(pk["cfg"]), B, C.byref(self.layout)), "wipa_decoder_layout")
        self.layout_key = tuple(getattr(self.layout, f) for f, _ in self.layout._fields_)
        with on_stream():
            self.blob = torch.empty(self.layout.total_bytes, dtype=torch.uint8, device=model.device)

    def _view(self, off: int, nbytes: int, dtype):
        return self.blob[off : off + nbytes].view(dtype)

    @property
    def tokens(self) -> torch.Tensor:
        lay = self.layout
        return self._view(lay.tokens, self.B * lay.ld_tok * 4, torch.int32).view(self.B, lay.ld_tok)

    @property
    def pos(self) -> torch.Tensor:
        return self._view(self.layout.pos, 4, torch.int32)

    @property
    def sum_logprobs(self) -> torch.Tensor:
        return self._view(self.layout.sum_logprobs, self.B * 4, torch.float32)

    @property
    def logits(self) -> torch.Tensor:
        lay = self.layout
        return self._view(lay.logits, self.B * lay.ld_logits * 4, torch.float32).view(self.B, lay.ld_logits)[:, : self.model.dims.n_vocab]

    def release(self):
        _lib.lib().wipa_decoder_release(ptr(self.blob))

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def _layout_of(pk, B: int) -> tuple:
    """the blob layout the packed configuration needs, as a comparable tuple"""
    lay = _lib.DecLayout()
    _lib.check(_lib.lib().wipa_decoder_layout(C.byref(pk["cfg"]), B, C.byref(lay)), "wipa_decoder_layout")
    return tuple(getattr(lay, f) for f, _ in lay._fields_)


def _packed_for(model, B: int, new_tokens: Optional[int]):
    """the weight tables + cfg for a decode of B clips x new_tokens positions: cross_attention="auto" picks the absorbed or
    the cached form from the measured table (Whisper.use_absorbed)"""
    return model.packed(absorbed=model.use_absorbed(B, new_tokens))


def _state_for(model, B: int, pk=None) -> DecoderState:
    """one decode blob per (model, library stream); re-made when the batch, the device or the LAYOUT changes.  The layout
    follows the configuration (dtype, cached / absorbed cross-attention, fp8 tables): a bf16 model that decoded with the
    absorbed form holds ONE copy of the features where the cached form needs K and V of every layer (24 x the bytes at
    whisper-small), so e.g. quantize_weights() after a first decode must not find the old blob."""
    pk = pk or model.packed()
    states = model.__dict__.setdefault("_dec_states", {})
    sid = (stream_id(), int(pk["cfg"].dec_cross_absorbed))  # one blob per library stream and cross-attention form
    st = states.get(sid)
    if st is not None and (st.B != B or st.blob.device != model.device or st.layout_key != _layout_of(pk, B)):
        stream().synchronize()
        st.release()
        st = None
    if st is None:
        st = DecoderState(model, B, pk)
        states[sid] = st
    return st


def _mask(model, ids: Sequence[int]) -> torch.Tensor:
    cache = model.__dict__.setdefault("_mask_cache", {})
    key = tuple(sorted(set(int(i) for i in ids)))
    m = cache.get(key)
    if m is None:
        host = torch.zeros(model.dims.n_vocab, dtype=torch.float32)
        if key:
            host[list(key)] = float("-inf")
        with on_stream() as s:
            m = host.to(model.device)
        s.synchronize()  # masks are shared by every library stream
        cache[key] = m
    return m


@dataclass
class GreedyHandle:
    state: "DecoderState"
    stream: torch.cuda.Stream
    n_init: int
    steps: int  # decoder steps enqueued (prompt positions included)
    keep: tuple = ()  # tensors that must outlive the enqueued work


def _use_prefill(n_init: int, total_steps: int) -> bool:
    """Batched prompt pass (wipa_decoder_prefill) unless WIPA_NO_PREFILL=1 asks for the step-by-step prompt."""
    return n_init >= 2 and total_steps >= n_init and os.environ.get("WIPA_NO_PREFILL") != "1"


def greedy_launch(model, audio_features: torch.Tensor, initial_tokens: Sequence[int], suppress_always: Sequence[int],
                  suppress_first: Sequence[int], eot: int, max_new_tokens: int, use_graph: bool = True) -> GreedyHandle:
    """Enqueue cross-KV projection + a FIXED number of decoder steps on the current library
    stream and return without synchronising (EOT rows are latched on the device, so running
    past the end of a row is harmless).  Pair with greedy_collect()."""
    L = _lib.lib()
    B = audio_features.shape[0]
    n_init = len(initial_tokens)
    max_new_tokens = min(max_new_tokens, model.dims.n_text_ctx - n_init)
    pk = _packed_for(model, B, max_new_tokens)
    m_always = _mask(model, suppress_always)
    m_first = _mask(model, list(suppress_always) + list(suppress_first))
    init = (C.c_int32 * n_init)(*[int(t) for t in initial_tokens])
    total = (n_init - 1) + max_new_tokens
    with on_stream() as s:
        st = _state_for(model, B, pk)
        feats = audio_features.to(device=model.device, dtype=model.dtype).contiguous()
        _lib.check(L.wipa_decoder_set_audio(C.byref(pk["cfg"]), pk["dec_tab"], ptr(feats), ptr(st.blob), st.blob.numel(), B, sptr(s)),
                   "wipa_decoder_set_audio")
        _lib.check(L.wipa_decoder_begin(C.byref(pk["cfg"]), ptr(st.blob), st.blob.numel(), B, init, n_init, sptr(s)), "wipa_decoder_begin")
        rest = total
        if _use_prefill(n_init, total):  # the prompt positions and the first new token in one batched pass
            _lib.check(L.wipa_decoder_prefill(C.byref(pk["cfg"]), pk["dec_tab"], ptr(st.blob), st.blob.numel(), B, n_init, eot, ptr(m_first),
                                              ptr(m_always), int(use_graph), sptr(s)), "wipa_decoder_prefill")
            rest = total - n_init
        _lib.check(L.wipa_decoder_run(C.byref(pk["cfg"]), pk["dec_tab"], ptr(st.blob), st.blob.numel(), B, n_init, eot, ptr(m_first),
                                      ptr(m_always), rest, int(use_graph), sptr(s)), "wipa_decoder_run")
    return GreedyHandle(st, s, n_init, total, (feats, m_always, m_first))


def greedy_collect(h: GreedyHandle) -> GreedyTokens:
    """Wait for a greedy_launch() and bring the token ids to the host."""
    with torch.cuda.stream(h.stream):
        toks = h.state.tokens[:, : h.steps + 1].cpu().numpy().astype(np.int64)
        slp = h.state.sum_logprobs.cpu().numpy().copy()
    h.stream.synchronize()
    return GreedyTokens(toks, h.steps - (h.n_init - 1), slp, h.state.logits)


def greedy_decode_tokens(model, audio_features: torch.Tensor, initial_tokens: Sequence[int], suppress_always: Sequence[int],
                         suppress_first: Sequence[int], eot: int, max_new_tokens: Optional[int] = None,
                         stop_on_eot: bool = True, use_graph: bool = True, check_every: int = 8) -> GreedyTokens:
    """DecodingTask._main_loop for temperature 0, n_group 1 (see module docstring).
    ``audio_features`` [B, 1500, d] in the model dtype."""
    L = _lib.lib()
    B = audio_features.shape[0]
    n_init = len(initial_tokens)
    if max_new_tokens is None:
        max_new_tokens = model.dims.n_text_ctx // 2
    max_new_tokens = min(max_new_tokens, model.dims.n_text_ctx - n_init)
    pk = _packed_for(model, B, max_new_tokens)
    st = _state_for(model, B, pk)
    m_always = _mask(model, suppress_always)
    m_first = _mask(model, list(suppress_always) + list(suppress_first))
    init = (C.c_int32 * n_init)(*[int(t) for t in initial_tokens])
    total = (n_init - 1) + max_new_tokens
    done_steps = 0
    with on_stream() as s:
        feats = audio_features.to(device=model.device, dtype=model.dtype).contiguous()
        _lib.check(L.wipa_decoder_set_audio(C.byref(pk["cfg"]), pk["dec_tab"], ptr(feats), ptr(st.blob), st.blob.numel(), B, sptr(s)),
                   "wipa_decoder_set_audio")
        _lib.check(L.wipa_decoder_begin(C.byref(pk["cfg"]), ptr(st.blob), st.blob.numel(), B, init, n_init, sptr(s)), "wipa_decoder_begin")
        if _use_prefill(n_init, total):
            _lib.check(L.wipa_decoder_prefill(C.byref(pk["cfg"]), pk["dec_tab"], ptr(st.blob), st.blob.numel(), B, n_init, eot, ptr(m_first),
                                              ptr(m_always), int(use_graph), sptr(s)), "wipa_decoder_prefill")
            done_steps = n_init
        while done_steps < total:
            n = min(check_every if stop_on_eot else total, total - done_steps)
            if done_steps == 0:
                n = min(total, n + n_init - 1)
            _lib.check(L.wipa_decoder_run(C.byref(pk["cfg"]), pk["dec_tab"], ptr(st.blob), st.blob.numel(), B, n_init, eot, ptr(m_first),
                                          ptr(m_always), n, int(use_graph), sptr(s)), "wipa_decoder_run")
            done_steps += n
            if stop_on_eot and done_steps >= n_init:
                last = st.tokens[:, done_steps].cpu()  # synchronises the library stream
                if bool((last == eot).all()):
                    break
        toks = st.tokens[:, : done_steps + 1].cpu().numpy().astype(np.int64)
        slp = st.sum_logprobs.cpu().numpy().copy()
        last_logits = st.logits
    n_steps = done_steps - (n_init - 1)
    if stop_on_eot:
        # the reference stops at the first step after which every row ends in EOT
        body = toks[:, n_init:]
        all_eot = (body == eot).all(axis=0)
        if all_eot.any():
            n_steps = int(np.argmax(all_eot)) + 1
            toks = toks[:, : n_init + n_steps]
    return GreedyTokens(toks, n_steps, slp, last_logits)


def forced_decode_logits(model, audio_features: torch.Tensor, tokens: np.ndarray, n_init: int, suppress_always: Sequence[int],
                         suppress_first: Sequence[int], eot: int, use_graph: bool = True):
    """The KV-cached decode-step path (prompt prefill + replayed step graph -- the kernels greedy_decode_tokens runs) driven
    along a GIVEN token history: after every step the greedy choice is read and then overwritten with ``tokens[:, p + 1]``.
    ``tokens`` [B, n_init + n_steps] int (host).  Returns (step logits [B, n_steps, V] f32 on the device, unfiltered, and the
    path's own greedy choices [B, n_steps] int64 on the host).  A measurement aid: with the history of an f32 reference it
    gives the logit error of a low-precision model at every step, independent of where its own greedy ids would part
    (DecodingTask._main_loop, transcribe_single.py:55)."""
    L = _lib.lib()
    tokens = np.asarray(tokens)
    B, total_len = tokens.shape
    assert B == audio_features.shape[0] and 2 <= n_init < total_len <= model.dims.n_text_ctx
    n_steps = total_len - n_init
    pk = _packed_for(model, B, n_steps)
    st = _state_for(model, B, pk)
    m_always = _mask(model, suppress_always)
    m_first = _mask(model, list(suppress_always) + list(suppress_first))
    init = (C.c_int32 * n_init)(*[int(t) for t in tokens[0, :n_init]])
    assert (tokens[:, :n_init] == tokens[0, :n_init]).all(), "one prompt for all rows"
    V = model.dims.n_vocab
    with on_stream() as s:
        forced = torch.from_numpy(tokens.astype(np.int32)).to(model.device)
        trace = torch.empty(B, n_steps, V, dtype=torch.float32, device=model.device)
        chosen = torch.empty(B, n_steps, dtype=torch.int32, device=model.device)
        feats = audio_features.to(device=model.device, dtype=model.dtype).contiguous()
        _lib.check(L.wipa_decoder_set_audio(C.byref(pk["cfg"]), pk["dec_tab"], ptr(feats), ptr(st.blob), st.blob.numel(), B, sptr(s)),
                   "wipa_decoder_set_audio")
        _lib.check(L.wipa_decoder_begin(C.byref(pk["cfg"]), ptr(st.blob), st.blob.numel(), B, init, n_init, sptr(s)), "wipa_decoder_begin")
        _lib.check(L.wipa_decoder_prefill(C.byref(pk["cfg"]), pk["dec_tab"], ptr(st.blob), st.blob.numel(), B, n_init, eot, ptr(m_first),
                                          ptr(m_always), int(use_graph), sptr(s)), "wipa_decoder_prefill")
        for i in range(n_steps):
            p = n_init + i  # the token position the last step has just filled
            trace[:, i].copy_(st.logits)
            chosen[:, i].copy_(st.tokens[:, p])
            st.tokens[:, p].copy_(forced[:, p])
            if i + 1 < n_steps:
                _lib.check(L.wipa_decoder_run(C.byref(pk["cfg"]), pk["dec_tab"], ptr(st.blob), st.blob.numel(), B, n_init, eot, ptr(m_first),
                                              ptr(m_always), 1, int(use_graph), sptr(s)), "wipa_decoder_run")
        out_chosen = chosen.cpu().numpy().astype(np.int64)
    return trace, out_chosen


def detect_language(model, audio_features: torch.Tensor, tokenizer: Tokenizer):
    """Whisper.detect_language (train_whisper_ipa.py:339 via language=None): one decoder pass on
    [sot]; every logit but the language tokens masked; argmax.  Returns (language token ids [B], probs)."""
    B = audio_features.shape[0]
    lang = list(tokenizer.all_language_tokens)
    not_lang = [i for i in range(model.dims.n_vocab) if i not in set(lang)]
    # one step on the prompt [sot]: with n_init = 1 the greedy kernel applies mask_first at position 0
    res = greedy_decode_tokens(model, audio_features, [tokenizer.sot], not_lang, [], eot=-1, max_new_tokens=1,
                               stop_on_eot=False)
    logits = res.last_logits[:, lang].float()
    probs = torch.softmax(logits, dim=-1).cpu().numpy()
    return res.tokens[:, 1], probs


def _suppress_lists(options: DecodingOptions, tok: Tokenizer):
    suppress = options.suppress_tokens
    if isinstance(suppress, str):
        suppress = [int(t) for t in suppress.split(",")]
    suppress = list(suppress or [])
    if -1 in suppress:
        suppress = [t for t in suppress if t >= 0] + list(tok.non_speech_tokens)
    suppress += [tok.transcribe, tok.translate, tok.sot, tok.sot_prev, tok.sot_lm, tok.no_speech]
    first = [tok.encode(" ")[0], tok.eot] if options.suppress_blank else []
    return sorted(set(suppress)), first


def decode(model, mel: torch.Tensor, options: DecodingOptions = DecodingOptions(), **kwargs):
    """mlx_whisper.decoding.decode: accepts a mel [B,3000,n_mels] / [3000,n_mels] or encoded
    features [B,1500,d] / [1500,d]; returns a DecodingResult or a list of them."""
    if kwargs:
        options = DecodingOptions(**{**options.__dict__, **kwargs})
    if options.beam_size or (options.best_of or 1) > 1 or options.temperature != 0.0:
        raise NotImplementedError("the reference only ever runs greedy decode (SURVEY.md section 0)")
    if not options.without_timestamps:
        raise NotImplementedError("timestamp rules are not on the reference's path (without_timestamps=True everywhere)")
    single = mel.dim() == 2
    if single:
        mel = mel[None]
    d = model.dims
    if tuple(mel.shape[-2:]) == (d.n_audio_ctx, d.n_audio_state):
        feats = mel
    else:
        feats = model.embed_audio(mel)
    if options.fp16 and model.dtype == torch.float32:
        # DecodingOptions.fp16=True (default; transcribe_single.py:49-52) casts the features to fp16
        # before cross-attention while the weights stay f32
        with on_stream():
            feats = feats.to(torch.float16).to(torch.float32)
    B = feats.shape[0]
    tok = get_tokenizer(model.is_multilingual, num_languages=model.num_languages, language=options.language or "en",
                        task=options.task)
    initial = list(tok.sot_sequence_including_notimestamps)
    languages = [options.language or "en"] * B
    lang_probs = [None] * B
    if options.language is None:
        # NOTE: one language per batch row in the reference; rows may differ, so decode per detected group
        lang_tokens, probs = detect_language(model, feats, tok)
        from .tokenizer import LANGUAGES
        languages = [LANGUAGES[int(t) - tok.sot - 1] for t in lang_tokens]
        lang_probs = [dict(zip(LANGUAGES[: tok.num_languages], p.tolist())) for p in probs]
    always, first = _suppress_lists(options, tok)
    sample_len = options.sample_len or d.n_text_ctx // 2
    results: List[Optional[DecodingResult]] = [None] * B
    for lang in sorted(set(languages)):
        rows = [i for i, l in enumerate(languages) if l == lang]
        init = list(initial)
        init[1] = tok.to_language_token(lang)
        sub = feats[rows] if len(rows) != B else feats
        g = greedy_decode_tokens(model, sub, init, always, first, tok.eot, max_new_tokens=sample_len)
        for j, i in enumerate(rows):
            row = g.tokens[j, len(init):].tolist()
            if tok.eot in row:
                row = row[: row.index(tok.eot)]
            text = tok.decode(row).strip()
            comp = len(text.encode("utf-8")) / max(len(zlib.compress(text.encode("utf-8"))), 1) if text else float("nan")
            results[i] = DecodingResult(audio_features=feats[i], language=lang, language_probs=lang_probs[i], tokens=row,
                                        text=text, avg_logprob=float(g.sum_logprobs[j]) / (len(row) + 1),
                                        temperature=options.temperature, compression_ratio=comp)
    return results[0] if single else results
