"""Several batches ("passes") of the mel -> encoder -> greedy decode path in flight on ONE GPU.

The reference's batch callers -- ``evaluate_model()`` (scripts/evaluate_model.py:127-232) and ``validate()``
(scripts/train_whisper_ipa.py:314-407) -- walk a list of clips and call load_audio -> pad_or_trim -> log_mel_spectrogram ->
model.encoder -> decode for one item at a time.  On an MI355X one such pass leaves most of the chip idle most of the time:
the encoder is MFMA-bound for a third of the pass and the decode loop is a chain of ~130 small dependent launches per step
for the rest.  ``transcribe_batches`` keeps ``passes_in_flight`` consecutive batches going, each on its own library stream
with its own workspaces, KV caches and captured step graphs, so the encoder of one batch runs beside the decode loops of the
others (whisper-small, 64 clips per batch, 64 new tokens: 72 ms per batch with 4 in flight against 110 ms one at a time;
DESIGN.md section 6).  Every clip's ids are the ids the one-at-a-time path gives (every kernel on the path is
batch-invariant and a pass never shares state with another; tests/test_gpu_model.py::test_transcribe_batches_*).

What the schedule sets, so that callers need not:
  * the library streams 0 .. passes_in_flight-1 (``runtime.use_stream``), one decode state per stream;
  * ``Whisper.cross_splits`` = 2 while >= 2 passes are in flight (half-chip streaming launches: two passes' cross-attention
    launches run side by side instead of queueing for all 256 CUs), the model's own setting otherwise; restored afterwards;
  * which hardware queue a pass lands on.  ROCm multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in
    creation order.  The library streams are the library's OWN (``wipa_stream_create``), the first eight created together
    before any carries work, so up to four passes in flight sit on distinct queues even with four (72.0 / 71.3 ms per pass
    against 72.1 / 71.8 / 71.3 with eight); torch's pool streams on four queues share them (86-88 ms), and a stream created
    while others carry work can land on a busy queue (85 ms on the decode loops).  The package still asks for 8 queues at
    import when nothing has initialised the GPU yet (``runtime.request_hw_queues``: headroom for other streams of the process,
    not speed), and this module warns only about what can still cost: torch's pool streams below 8 queues
    (``WIPA_OWN_STREAMS=0``), or more passes in flight than streams created together.

Decode groups (``decode_group=G``, default 1 = off): the batches still go through log-mel and the encoder one by one, but the
rows of G consecutive batches then decode as ONE chain of G x B rows -- fewer, fatter dependent-launch chains on the GPU (four
chains active at once cost every chain ~2.6 x its launch gaps, profiles/r05_decode_gaps.txt) and the decoder weights streamed
once per step for the group.  A clip's ids do not depend on the rows it decodes beside (every kernel is batch-invariant), so
the results are those of G = 1; what changes is WHEN: the first batch of a group waits for the group's last encoder.
``passes_in_flight`` then counts groups.

Early stop: the reference's loop ends at the first step after which every row has emitted EOT.  Here the steps of a pass are
enqueued in chunks of ``check_every``; after each chunk the last token column is copied to pinned host memory behind an
event, and the host -- which is never blocked on one pass while another has room for work -- stops enqueueing for a pass once
a probe shows every row at EOT.  Rows are EOT-latched on the device, so the extra steps of the chunk already in flight change
neither ids nor log-probabilities.
"""
from __future__ import annotations

import ctypes as C
import queue
import threading
import warnings
import zlib
from collections import deque
from dataclasses import dataclass, field
from typing import Deque, Iterable, Iterator, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from . import audio as A
from .decoding import (DecodingOptions, DecodingResult, _mask, _packed_for, _state_for, _suppress_lists, _use_prefill)
import os

from .runtime import OWN_STREAM_COUNT, hw_queues, ptr, sptr, use_stream
from .tokenizer import LANGUAGES, get_tokenizer

PIPELINE_HW_QUEUES = 8          # what 4 passes in flight want (one queue per pass is not enough: copies and graph launches share them)
PIPELINE_CROSS_SPLITS = 2       # Whisper.cross_splits while >= 2 passes are in flight
CHUNKS_AHEAD = 2                # chunks of decode steps kept enqueued per pass while its EOT probes are outstanding


@dataclass
class PassResult:
    """One batch's outcome.  ``tokens`` [B, n_init + n_steps] int64 on the host (prompt included, EOT-latched rows);
    ``results`` builds the reference's ``DecodingResult`` list (text through the tokenizer) on first use."""
    tokens: np.ndarray
    n_init: int
    n_steps: int
    sum_logprobs: np.ndarray
    languages: List[str]
    language_probs: List[Optional[dict]]
    audio_features: torch.Tensor
    index: int = 0                       # position of the batch in the input sequence
    _tok: object = None
    _eot: int = 0
    _temperature: float = 0.0
    _results: Optional[List[DecodingResult]] = field(default=None, repr=False)

    def rows(self) -> List[List[int]]:
        """new tokens of every clip, cut at the first EOT (DecodingTask.run: tokens[sample_begin : first eot])"""
        out = []
        for r in self.tokens[:, self.n_init:]:
            r = r.tolist()
            out.append(r[: r.index(self._eot)] if self._eot in r else r)
        return out

    @property
    def results(self) -> List[DecodingResult]:
        if self._results is None:
            res = []
            for i, row in enumerate(self.rows()):
                text = self._tok.decode(row).strip()
                comp = len(text.encode("utf-8")) / max(len(zlib.compress(text.encode("utf-8"))), 1) if text else float("nan")
                res.append(DecodingResult(audio_features=self.audio_features[i], language=self.languages[i],
                                          language_probs=self.language_probs[i], tokens=row, text=text,
                                          avg_logprob=float(self.sum_logprobs[i]) / (len(row) + 1), temperature=self._temperature,
                                          compression_ratio=comp))
            self._results = res
        return self._results

    @property
    def texts(self) -> List[str]:
        return [r.text for r in self.results]


@dataclass
class _Probe:
    event: torch.cuda.Event
    column: torch.Tensor  # pinned host copy of the token column after ``steps`` decoder steps
    steps: int


@dataclass
class _Pass:
    index: int
    slot: int
    stream: torch.cuda.Stream
    state: object
    pk: dict
    B: int
    n_init: int
    total: int                 # decoder steps of the full length (prompt positions included)
    enqueued: int              # decoder steps enqueued so far
    masks: tuple
    feats: torch.Tensor
    eot: int
    probes: Deque[_Probe] = field(default_factory=deque)
    stop_at: Optional[int] = None     # steps after which a probe saw every row at EOT
    lang_tok: Optional[torch.Tensor] = None
    lang_logits: Optional[torch.Tensor] = None
    keep: tuple = ()
    sizes: tuple = ()      # clips of every batch of the decode group, in input order (sum = B)
    indices: tuple = ()    # their positions in the input sequence

    @property
    def may_enqueue(self) -> bool:
        return self.stop_at is None and self.enqueued < self.total


def _prefetched(it: Iterable, depth: int) -> Iterator:
    """the caller's (host-only) batch iterator on a helper thread, ``depth`` batches ahead: reading and decoding audio files must
    not keep the thread that feeds the GPU away from its streams"""
    q: "queue.Queue" = queue.Queue(maxsize=max(1, depth))
    END = object()

    def work():
        try:
            for item in it:
                q.put((item, None))
            q.put((END, None))
        except BaseException as e:  # surfaces in the consumer
            q.put((END, e))

    th = threading.Thread(target=work, daemon=True, name="wipa-batch-prefetch")
    th.start()
    while True:
        item, err = q.get()
        if item is END:
            if err is not None:
                raise err
            return
        yield item


class TranscribePipeline:
    """The scheduler behind ``transcribe_batches``; usable directly when batches arrive one by one:
    ``with TranscribePipeline(model, options) as p: p.submit(batch) ...; for r in p.drain(): ...``.
    ``submit`` returns the results that had to be collected to make room (possibly none), in input order.
    One pipeline per model at a time, driven from ONE thread: the passes share the model's per-stream workspaces and decode
    states (keyed by library stream 0 .. passes_in_flight - 1), and ``cross_splits`` is the model's while the block is open."""

    def __init__(self, model, options: Optional[DecodingOptions] = None, passes_in_flight: int = 4, *,
                 max_new_tokens: Optional[int] = None, stop_on_eot: bool = True, check_every: int = 8,
                 cross_splits: Optional[int] = None, use_graph: bool = True, decode_group: int = 1):
        options = options or DecodingOptions(language="en", without_timestamps=True)
        if options.beam_size or (options.best_of or 1) > 1 or options.temperature != 0.0:
            raise NotImplementedError("the reference only ever runs greedy decode (SURVEY.md section 0)")
        if not options.without_timestamps:
            raise NotImplementedError("timestamp rules are not on the reference's path (without_timestamps=True everywhere)")
        if passes_in_flight < 1:
            raise _lib.WipaError(f"passes_in_flight must be >= 1, got {passes_in_flight}")
        if decode_group < 1:
            raise _lib.WipaError(f"decode_group must be >= 1, got {decode_group}")
        self.G = int(decode_group)
        self.pending: list = []   # (batch, index) waiting for their decode group to fill
        self.model, self.options, self.P = model, options, int(passes_in_flight)
        self.stop_on_eot, self.check_every, self.use_graph = bool(stop_on_eot), max(1, int(check_every)), bool(use_graph)
        d = model.dims
        self.tok = get_tokenizer(model.is_multilingual, num_languages=model.num_languages, language=options.language or "en",
                                 task=options.task)
        self.initial = list(self.tok.sot_sequence_including_notimestamps)
        self.always, self.first = _suppress_lists(options, self.tok)
        n_new = max_new_tokens if max_new_tokens is not None else (options.sample_len or d.n_text_ctx // 2)
        self.max_new = min(int(n_new), d.n_text_ctx - len(self.initial))
        self.lang_ids = list(self.tok.all_language_tokens) if options.language is None else None
        if self.lang_ids is not None:
            keep = set(self.lang_ids)
            self.not_lang = [i for i in range(d.n_vocab) if i not in keep]
        # the setting for several passes in flight; an explicit cross_splits wins
        self._splits = (PIPELINE_CROSS_SPLITS if self.P >= 2 else model.cross_splits) if cross_splits is None else int(cross_splits)
        self._splits_before: Optional[int] = None
        self.inflight: Deque[_Pass] = deque()
        self.submitted = 0   # batches handed over
        self.launched = 0    # passes (batches, or decode groups) enqueued
        self.hw_queues = hw_queues()
        own = os.environ.get("WIPA_OWN_STREAMS", "1") == "1"
        if self.P > OWN_STREAM_COUNT or (not own and self.P >= 2 and self.hw_queues < PIPELINE_HW_QUEUES):
            # with the library's own streams (the default) four hardware queues serve four passes as well as eight; what is left to
            # warn about: more passes than streams created together, or torch's pool streams on fewer than 8 queues (they share)
            warnings.warn(f"whisper_ipa_amd: {self.P} passes in flight on {self.hw_queues} hardware queues"
                          + (" with torch's pool streams (WIPA_OWN_STREAMS=0): they share queues below 8 (measured: 86-88 ms per pass "
                             "against 72)" if not own else f", more than the {OWN_STREAM_COUNT} library streams that are created together"),
                          RuntimeWarning, stacklevel=3)

    # ---- context: the model's streaming-launch setting belongs to the schedule while it runs
    def __enter__(self):
        self._splits_before = self.model.cross_splits
        if self.model.cross_splits != self._splits:
            self.model.cross_splits = self._splits
        return self

    def __exit__(self, *exc):
        try:
            if exc[0] is None:
                assert not self.inflight and not self.pending, "TranscribePipeline closed with batches pending or in flight: drain() first"
            else:  # an error: let the enqueued work finish before the states are reused; drop what was never launched
                for p in self.inflight:
                    p.stream.synchronize()
                self.inflight.clear()
                self.pending.clear()
        finally:
            if self._splits_before is not None and self.model.cross_splits != self._splits_before:
                self.model.cross_splits = self._splits_before
            self._splits_before = None

    # ---- one pass
    def _features(self, batch) -> torch.Tensor:
        """audio [B, samples] (padded / cut to 30 s) | mel [B, 3000, n_mels] | features [B, 1500, d] -> features, enqueued on the
        current library stream"""
        m, d = self.model, self.model.dims
        if isinstance(batch, dict):
            batch = batch.get("audio_features", batch.get("mel_features", batch.get("audio")))
        t = torch.as_tensor(batch)
        if t.dim() == 3 and tuple(t.shape[-2:]) == (d.n_audio_ctx, d.n_audio_state):
            feats = t.to(device=m.device, non_blocking=True)
        elif t.dim() == 3:
            feats = m.embed_audio(t.to(device=m.device, non_blocking=True))
        elif t.dim() == 2:
            a = A.pad_or_trim(t.to(device=m.device, dtype=torch.float32, non_blocking=True))
            feats = m.encode_padded(A.log_mel_padded(a, d.n_mels, m.dtype), a.shape[0])
        else:
            raise _lib.WipaError(f"transcribe_batches: a batch is audio [B, samples], mel [B, 3000, n_mels] or features "
                                 f"[B, {d.n_audio_ctx}, {d.n_audio_state}]; got shape {tuple(t.shape)}")
        if self.options.fp16 and m.dtype == torch.float32:
            feats = feats.to(torch.float16).to(torch.float32)  # DecodingOptions.fp16, as decoding.decode
        return feats.to(m.dtype).contiguous()

    def _run(self, p: _Pass, n: int) -> None:
        L = _lib.lib()
        st = p.state
        _lib.check(L.wipa_decoder_run(C.byref(p.pk["cfg"]), p.pk["dec_tab"], ptr(st.blob), st.blob.numel(), p.B, p.n_init, p.eot,
                                      ptr(p.masks[1]), ptr(p.masks[0]), n, int(self.use_graph), sptr(p.stream)), "wipa_decoder_run")
        p.enqueued += n

    def _enqueue_chunks(self, p: _Pass) -> None:
        """keep CHUNKS_AHEAD chunks (each followed by an EOT probe) enqueued on the pass's stream"""
        with use_stream(p.slot):
            while p.may_enqueue and len(p.probes) < CHUNKS_AHEAD:
                self._run(p, min(self.check_every, p.total - p.enqueued))
                col = torch.empty(p.B, dtype=torch.int32, pin_memory=True)
                col.copy_(p.state.tokens[:, p.enqueued], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(p.stream)
                p.probes.append(_Probe(ev, col, p.enqueued))

    def _launch(self, group, slot: int) -> _Pass:
        """``group``: [(batch, index), ...] -- one batch, or the batches of a decode group"""
        L = _lib.lib()
        m = self.model
        cur = torch.cuda.current_stream()
        with use_stream(slot) as s:
            if cur != s:
                s.wait_stream(cur)  # the batch may have been produced on the caller's stream
            parts = [self._features(b) for b, _ in group]  # log-mel + encoder per batch, as without groups
            sizes = tuple(int(f.shape[0]) for f in parts)
            feats = parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)  # the group's rows decode as one chain
            del parts
            B = feats.shape[0]
            n_init = len(self.initial)
            total = (n_init - 1) + self.max_new
            pk = _packed_for(m, B, self.max_new)
            st = _state_for(m, B, pk)
            m_always = _mask(m, self.always)
            m_first = _mask(m, list(self.always) + list(self.first))
            keep = [feats, m_always, m_first]
            cfg, tab, blob, nb = C.byref(pk["cfg"]), pk["dec_tab"], ptr(st.blob), st.blob.numel()
            _lib.check(L.wipa_decoder_set_audio(cfg, tab, ptr(feats), blob, nb, B, sptr(s)), "wipa_decoder_set_audio")
            lang_tok = lang_logits = None
            if self.lang_ids is not None:
                # Whisper.detect_language (train_whisper_ipa.py:339 via language=None): one decoder pass on [sot], every logit but
                # the language tokens masked, argmax -- as decoding.detect_language, but the winners stay on the device and go
                # straight into column 1 of the prompt: rows of one batch may carry different language tokens, no host round trip
                m_lang = _mask(m, self.not_lang)
                keep.append(m_lang)
                sot = (C.c_int32 * 1)(int(self.tok.sot))
                _lib.check(L.wipa_decoder_begin(cfg, blob, nb, B, sot, 1, sptr(s)), "wipa_decoder_begin")
                _lib.check(L.wipa_decoder_run(cfg, tab, blob, nb, B, 1, -1, ptr(m_lang), ptr(m_lang), 1, int(self.use_graph), sptr(s)),
                           "wipa_decoder_run")
                lang_tok = st.tokens[:, 1].clone()
                lang_logits = st.logits[:, self.lang_ids].float().clone()
            init = (C.c_int32 * n_init)(*[int(t) for t in self.initial])
            _lib.check(L.wipa_decoder_begin(cfg, blob, nb, B, init, n_init, sptr(s)), "wipa_decoder_begin")
            if lang_tok is not None:
                st.tokens[:, 1].copy_(lang_tok)
            p = _Pass(group[0][1], slot, s, st, pk, B, n_init, total, 0, (m_always, m_first), feats, int(self.tok.eot),
                      lang_tok=lang_tok, lang_logits=lang_logits, keep=tuple(keep), sizes=sizes, indices=tuple(i for _, i in group))
            if _use_prefill(n_init, total):  # the prompt positions and the first new token in one batched pass
                _lib.check(L.wipa_decoder_prefill(cfg, tab, blob, nb, B, n_init, p.eot, ptr(m_first), ptr(m_always), int(self.use_graph),
                                                  sptr(s)), "wipa_decoder_prefill")
                p.enqueued = n_init
            if not self.stop_on_eot:
                if p.enqueued < total:
                    self._run(p, total - p.enqueued)  # the whole fixed-length decode in one call (bench.py's timed passes)
            elif p.enqueued == 0:
                self._run(p, min(total, n_init - 1 + self.check_every))  # prompt positions step by step + the first chunk
        if self.stop_on_eot:
            self._enqueue_chunks(p)
        return p

    def _advance(self, p: _Pass) -> None:
        """consume the pass's EOT probes that have completed (never blocks) and top its chunks up"""
        while p.probes and p.probes[0].event.query():
            pr = p.probes.popleft()
            if p.stop_at is None and bool((pr.column == p.eot).all()):
                p.stop_at = pr.steps
        self._enqueue_chunks(p)

    def _pump(self) -> None:
        for p in self.inflight:
            self._advance(p)

    def _collect(self, p: _Pass) -> List[PassResult]:
        """``p`` has left ``self.inflight``: drive it to its end (the younger passes keep their chunks topped up meanwhile)"""
        while self.stop_on_eot:
            self._advance(p)
            self._pump()
            if not p.probes:
                break
            p.probes[0].event.synchronize()  # the oldest pass's next probe; the others keep CHUNKS_AHEAD chunks meanwhile
        with torch.cuda.stream(p.stream):
            toks = p.state.tokens[:, : p.enqueued + 1].cpu().numpy().astype(np.int64)
            slp = p.state.sum_logprobs.cpu().numpy().copy()
            lang_tok = p.lang_tok.cpu().numpy() if p.lang_tok is not None else None
            lang_logits = p.lang_logits.cpu() if p.lang_logits is not None else None
        p.stream.synchronize()
        languages_all = [self.options.language or "en"] * p.B
        probs_all: List[Optional[dict]] = [None] * p.B
        if lang_tok is not None:
            languages_all = [LANGUAGES[int(t) - self.tok.sot - 1] for t in lang_tok]
            pr = torch.softmax(lang_logits, dim=-1).numpy()
            probs_all = [dict(zip(LANGUAGES[: self.tok.num_languages], row.tolist())) for row in pr]
        out, r0 = [], 0
        for size, index in zip(p.sizes, p.indices):  # every batch of a decode group gets the result it gets alone
            t = toks[r0:r0 + size]
            n_steps = p.enqueued - (p.n_init - 1)
            if self.stop_on_eot:
                # the reference stops at the first step after which every row ends in EOT (as decoding.greedy_decode_tokens): per
                # BATCH -- rows are EOT-latched, so the steps a batch rides along with its group change nothing
                all_eot = (t[:, p.n_init:] == p.eot).all(axis=0)
                if all_eot.any():
                    n_steps = int(np.argmax(all_eot)) + 1
                    t = t[:, : p.n_init + n_steps]
            out.append(PassResult(t, p.n_init, n_steps, slp[r0:r0 + size], languages_all[r0:r0 + size], probs_all[r0:r0 + size],
                                  p.feats[r0:r0 + size], index=index, _tok=self.tok, _eot=p.eot, _temperature=self.options.temperature))
            r0 += size
        return out

    # ---- the schedule
    def submit(self, batch) -> List[PassResult]:
        """enqueue one batch; when every slot is taken the OLDEST pass is collected first (its stream set is the one reused, so a
        pass's work is never ordered behind a younger pass).  Returns what was collected, in input order (possibly nothing).
        With ``decode_group`` G > 1 the batch waits until G are there (``drain`` flushes a partial group)."""
        if self._splits_before is None:
            raise _lib.WipaError("TranscribePipeline.submit outside its ``with`` block")
        self.pending.append((batch, self.submitted))
        self.submitted += 1
        if len(self.pending) < self.G:
            return []
        return self._launch_pending()

    def _launch_pending(self) -> List[PassResult]:
        done: List[PassResult] = []
        if len(self.inflight) == self.P:
            done += self._collect(self.inflight.popleft())
        slot = self.launched % self.P
        group, self.pending = self.pending, []
        self.inflight.append(self._launch(group, slot))
        self.launched += 1
        if self.stop_on_eot:
            self._pump()
        return done

    def drain(self) -> Iterator[PassResult]:
        if self.pending:
            for r in self._launch_pending():
                yield r
        while self.inflight:
            for r in self._collect(self.inflight.popleft()):
                yield r


def transcribe_batches(model, batches: Iterable, options: Optional[DecodingOptions] = None, passes_in_flight: int = 4, *,
                       max_new_tokens: Optional[int] = None, stop_on_eot: bool = True, check_every: int = 8,
                       cross_splits: Optional[int] = None, prefetch: int = 0, decode_group: int = 1) -> Iterator[PassResult]:
    """Transcribe a sequence of batches with ``passes_in_flight`` of them in flight; yields one ``PassResult`` per batch, in
    input order (``.results``: the reference's DecodingResult list; ``.texts``; ``.tokens``).

    ``batches``: an iterable of audio [B, samples] (f32, host or device; padded / cut to 30 s), mel [B, 3000, n_mels] or encoder
    features [B, 1500, d] -- or dicts carrying one of ``audio_features`` / ``mel_features`` / ``audio`` (IPADataset.get_batch).
    It is consumed lazily, one batch per free slot.  ``prefetch`` > 0 iterates it on a helper thread that many batches ahead
    (for iterables that only do HOST work -- reading and resampling audio files; anything that enqueues GPU work should stay on
    the calling thread).
    ``options``: as ``decode`` (language=None detects the language per clip on the device).  ``max_new_tokens`` (default
    ``options.sample_len`` or n_text_ctx // 2 = 224) and ``stop_on_eot`` as ``greedy_decode_tokens``; ``stop_on_eot=False``
    enqueues a pass's whole fixed-length decode at once.
    ``passes_in_flight=1`` is the serial schedule (one batch at a time, the model's own ``cross_splits``).
    ``decode_group=G``: G consecutive batches decode as one chain of rows (module docstring); results unchanged, still one per batch."""
    it = _prefetched(batches, prefetch) if prefetch > 0 else iter(batches)
    with TranscribePipeline(model, options, passes_in_flight, max_new_tokens=max_new_tokens, stop_on_eot=stop_on_eot,
                            check_every=check_every, cross_splits=cross_splits, decode_group=decode_group) as pipe:
        for batch in it:
            for r in pipe.submit(batch):
                yield r
        for r in pipe.drain():
            yield r
