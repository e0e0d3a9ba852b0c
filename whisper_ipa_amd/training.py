"""Decoder-only fine-tune step (f32) with the semantics of the reference's ``train_step``
(scripts/train_whisper_ipa.py:266-311): encoder frozen (:187), teacher forcing on
``tokens[:, :-1]`` against ``tokens[:, 1:]`` (:228-232), masked CE with the batch-GLOBAL valid count
(:242-261), gradients w.r.t. every ``decoder.*`` tensor (tied embedding included), PER-TENSOR clip
(:287-303), mlx-style AdamW without bias correction (:513).

mlx's ``nn.value_and_grad`` is replaced by an explicit forward that saves activations and a
hand-written backward; every arithmetic step is a libwipa kernel (GEMMs through wipa_gemm on
operands transposed by wipa_transpose, flash-style attention backward, LayerNorm/GELU/CE backward,
fused clip+AdamW over one flat parameter buffer).  torch only owns the buffers.

Data parallel (one process per GPU): the (sum CE, valid count) pair is all-reduced BEFORE the
backward so the normalisation is global, the flat gradient buffer is all-reduced in buckets AFTER
it, and the per-tensor clip runs on the reduced gradients -- single-process semantics are kept.
"""
from __future__ import annotations


import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib, ops, parallel
from .runtime import on_stream, ptr, sptr
from .whisper import Whisper, parameter_names

QK_SCALE = 64 ** -0.25
OPT_CHUNK = 4096


def _ceil(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class FrozenFeatureCache:
    """HBM-resident cache of the FROZEN encoder's output, keyed by clip (dataset index).

    The reference recomputes ``model.embed_audio(mel)`` in every training step although the encoder is frozen
    (scripts/train_whisper_ipa.py:187,223): its output is a pure function of the clip.  whisper-small
    features are 1500 x 768 f32 = 4.6 MB per clip, so the 7 000 training clips of data/v2_filtered are 32 GB of a 288 GB
    HBM: keep them.  A clip's features are computed once, by the same kernels, and are bit-identical whatever batch they are
    computed in (tests/test_gpu_training.py), so training with the cache gives bit-identical losses and parameters.
    Clips beyond ``max_clips`` are simply recomputed every time (spill to recompute, no eviction).  The entries are
    functions of the encoder weights: the model counts every change of an ``encoder.*`` tensor (load_weights / update /
    set_dtype / quantize_weights -> ``Whisper._enc_generation``) and the cache drops itself when the count moved, so stale
    features cannot survive an encoder update; DecoderTrainer itself never touches encoder tensors."""

    def __init__(self, model: Whisper, max_clips: int):
        d = model.dims
        self.model, self.max_clips = model, int(max_clips)
        self.row_shape = (d.n_audio_ctx, d.n_audio_state)
        self.slots: Dict[int, int] = {}
        self.store: Optional[torch.Tensor] = None  # [max_clips, 1500, d] f32, allocated on first use
        self._pinned = None  # ring of [pinned int64 [2, n], event of the last upload]: host staging of the gather indices
        self._ring = 0
        self.hits = self.misses = 0
        self._enc_generation = model._enc_generation

    def _drop_if_encoder_changed(self) -> None:
        if self._enc_generation != self.model._enc_generation:
            self.clear()
            self._enc_generation = self.model._enc_generation

    @staticmethod
    def clips_that_fit(model: Whisper, fraction_of_free: float = 0.5) -> int:
        free, _ = torch.cuda.mem_get_info(model.device)
        return int(free * fraction_of_free) // (model.dims.n_audio_ctx * model.dims.n_audio_state * 4)

    def clear(self) -> None:
        self.slots.clear()
        self.hits = self.misses = 0

    def missing(self, keys) -> List[bool]:
        """per key: True when the clip's features are not cached (its audio / mel is needed)"""
        self._drop_if_encoder_changed()
        return [int(k) not in self.slots for k in keys]

    def assemble(self, keys, mel_of_missing: Optional[torch.Tensor]) -> torch.Tensor:
        """features [B, 1500, d] f32 for ``keys``; ``mel_of_missing`` [n_missing, 3000, n_mels] holds the mel of the keys
        for which missing() is True, in order (None when there are none)."""
        self._drop_if_encoder_changed()
        keys = [int(k) for k in keys]
        miss_pos = [i for i, k in enumerate(keys) if k not in self.slots]
        n_miss = 0 if mel_of_missing is None else mel_of_missing.shape[0]
        if n_miss != len(miss_pos):
            raise _lib.WipaError(f"FrozenFeatureCache: {len(miss_pos)} clips are not cached but the mel of {n_miss} was given")
        with on_stream():
            out = torch.empty(len(keys), *self.row_shape, dtype=torch.float32, device=self.model.device)
            if miss_pos:
                fresh = self.model.embed_audio(mel_of_missing)
                for j, i in enumerate(miss_pos):
                    out[i].copy_(fresh[j])
                    k = keys[i]
                    if k not in self.slots and len(self.slots) < self.max_clips:
                        if self.store is None:
                            self.store = torch.empty(self.max_clips, *self.row_shape, dtype=torch.float32, device=self.model.device)
                        self.slots[k] = len(self.slots)
                        self.store[self.slots[k]].copy_(fresh[j])
            miss = set(miss_pos)
            hit_pos = [i for i in range(len(keys)) if i not in miss]
            if hit_pos:
                # the two index vectors go up through a pinned staging buffer with non-blocking copies: a torch.tensor(...,
                # device=...) from pageable memory makes the host wait for the stream to drain, i.e. for the whole previous step,
                # and the GPU then idles while the host enqueues this one (7 ms per step at 32 x 64)
                n = len(hit_pos)
                # a ring of four staging buffers, each guarded by the event of its last upload: the host may run several steps
                # ahead of the GPU, and a buffer must not be rewritten before its copy has been executed
                if self._pinned is None or self._pinned[0][0].shape[1] < n:
                    self._pinned = [[torch.empty(2, max(n, 64), dtype=torch.int64).pin_memory(), None] for _ in range(4)]
                    self._ring = 0
                buf = self._pinned[self._ring]
                self._ring = (self._ring + 1) % len(self._pinned)
                if buf[1] is not None:
                    buf[1].synchronize()
                buf[0][0, :n] = torch.tensor([self.slots[keys[i]] for i in hit_pos], dtype=torch.int64)
                buf[0][1, :n] = torch.tensor(hit_pos, dtype=torch.int64)
                dev = buf[0][:, :n].to(self.model.device, non_blocking=True)
                buf[1] = torch.cuda.Event()
                buf[1].record(torch.cuda.current_stream(self.model.device))
                if n == len(keys) and hit_pos == list(range(n)):
                    torch.index_select(self.store, 0, dev[0], out=out)
                else:
                    out.index_copy_(0, dev[1], self.store.index_select(0, dev[0]))
        self.hits += len(hit_pos)
        self.misses += len(miss_pos)
        return out


CLIP_SCOPES = ("reference", "all")


def clip_reaches(name: str, scope: str = "reference") -> bool:
    """Does the reference's ``clip_grad_dict`` (train_whisper_ipa.py:287-303) reach the gradient ``name``?  It walks dict
    values only: a tensor below a list (``decoder.blocks.{i}....``; mlx_whisper keeps the blocks in a Python list, and the
    reference's flatten_params :50-53 has the matching list branch) is handed back as it came (:299-300)."""
    if scope == "all":
        return True
    return not any(part.isdigit() for part in name.split("."))


class DecoderTrainer:
    def __init__(self, model: Whisper, lr: float = 1e-5, max_grad_norm: float = 1.0, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.01, f32_split: Optional[bool] = None, clip_scope: str = "reference"):
        """``f32_split`` (default: the model's setting -- on unless the model was built with f32_split=False): split-bf16 products in the large GEMMs of the step.
        ``clip_scope``: which tensors the per-tensor clip reaches.  ``"reference"`` (default) is ``clip_grad_dict`` as the
        reference wrote it (train_whisper_ipa.py:287-303): it recurses through dict values only, so the ``decoder.blocks``
        LIST is passed through (:299-300) and only ``token_embedding.weight``, ``positional_embedding`` and ``ln.*`` are
        clipped; ``"all"`` clips every decoder tensor by its own norm (what that function's docstring intends)."""
        if clip_scope not in CLIP_SCOPES:
            raise _lib.WipaError(f"DecoderTrainer: clip_scope {clip_scope!r}: one of {CLIP_SCOPES}")
        self.clip_scope = clip_scope
        if model.dtype != torch.float32:
            raise _lib.WipaError("DecoderTrainer: the fine-tune step runs in float32 (reference: set_dtype(mx.float32))")
        self.model, self.lr, self.max_grad_norm = model, lr, max_grad_norm
        self.b1, self.b2 = betas
        self.eps, self.wd = eps, weight_decay
        self.L = _lib.lib()
        self.f32_split = model.f32_split if f32_split is None else bool(f32_split)
        d = model.dims
        self.names = [n for n in parameter_names(d) if n.startswith("decoder.")]
        P = model.flat_parameters()
        sizes = [P[n].numel() for n in self.names]
        assert all(s % 4 == 0 for s in sizes)
        offs, total = [], 0
        for s in sizes:
            offs.append(total)
            total += s
        self.offsets = dict(zip(self.names, offs))
        self.shapes = {n: tuple(P[n].shape) for n in self.names}
        self.n_params = total
        # contiguous gradient segments in flat order: [embeddings][block 0]...[block L-1][final ln]; the
        # backward finishes them from the back, so each can be all-reduced while earlier blocks still compute
        starts = [offs[self.names.index(f"decoder.blocks.{l}.attn.query.weight")] for l in range(d.n_text_layer)]
        tail = offs[self.names.index("decoder.ln.weight")]
        self.block_ranges = [(starts[l], starts[l + 1] if l + 1 < d.n_text_layer else tail) for l in range(d.n_text_layer)]
        self.head_range, self.tail_range = (0, starts[0]), (tail, total)
        with on_stream():
            dev = model.device
            self.flat_p = torch.empty(total, dtype=torch.float32, device=dev)
            self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
            self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
            self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
            for n in self.names:
                self.p(n).copy_(P[n])
                model._params[n] = self.p(n)  # the model now reads the optimiser's buffer directly
            # chunk tables for the flat multi-tensor optimiser
            ch_off, ch_len, ch_seg, seg_first = [], [], [], [0]
            for si, (o, s) in enumerate(zip(offs, sizes)):
                for c0 in range(0, s, OPT_CHUNK):
                    ch_off.append(o + c0)
                    ch_len.append(min(OPT_CHUNK, s - c0))
                    ch_seg.append(si)
                seg_first.append(len(ch_off))
            self.n_chunks, self.n_seg = len(ch_off), len(sizes)
            self.ch_off = torch.tensor(ch_off, dtype=torch.int64, device=dev)
            self.ch_len = torch.tensor(ch_len, dtype=torch.int32, device=dev)
            self.ch_seg = torch.tensor(ch_seg, dtype=torch.int32, device=dev)
            self.seg_first = torch.tensor(seg_first, dtype=torch.int32, device=dev)
            self.partial = torch.empty(self.n_chunks, dtype=torch.float32, device=dev)
            self.coef = torch.empty(self.n_seg, dtype=torch.float32, device=dev)
            self.norms = torch.empty(self.n_seg, dtype=torch.float32, device=dev)
            self.seg_clip = torch.tensor([int(clip_reaches(n, clip_scope)) for n in self.names], dtype=torch.int32, device=dev)
        model._invalidate()
        self.step_count = 0
        self._colsum_ws = None  # partial sums of the chunked bias-gradient reduction (wipa_colsum)
        self._dw_slabs = None   # split-K partial weight gradients (wipa_gemm k_slices + wipa_sum_slabs)
        self._mm_slabs = None   # split-K partial outputs of the token-row GEMMs (_mm)
        self.feature_cache: Optional[FrozenFeatureCache] = None  # enable_feature_cache()
        self._leaves = None     # autograd leaves over flat_p (leaves())
        model._trainer = self   # Whisper.logits differentiates through this trainer under torch.enable_grad()

    def feature_cache_capacity_default(self) -> int:
        return FrozenFeatureCache.clips_that_fit(self.model)

    def enable_feature_cache(self, max_clips: Optional[int] = None) -> "FrozenFeatureCache":
        """Keep the frozen encoder's output per clip in HBM (FrozenFeatureCache); ``max_clips`` defaults to what half of the
        free device memory holds."""
        n = FrozenFeatureCache.clips_that_fit(self.model) if max_clips is None else int(max_clips)
        self.feature_cache = FrozenFeatureCache(self.model, max(0, n))
        return self.feature_cache

    # ---- views into the flat buffers
    def p(self, name: str) -> torch.Tensor:
        o = self.offsets[name]
        return self.flat_p[o:o + _numel(self.shapes[name])].view(self.shapes[name])

    def g(self, name: str) -> torch.Tensor:
        o = self.offsets[name]
        return self.flat_g[o:o + _numel(self.shapes[name])].view(self.shapes[name])

    def grads(self) -> Dict[str, torch.Tensor]:
        return {n: self.g(n) for n in self.names}

    # ---- kernel helpers (all on the library stream)
    def _gemm(self, *a, **kw):
        return ops.gemm(*a, f32_split=self.f32_split, **kw)

    def _lin(self, x, M, K, W, N, bias=None, scale=None, residual=None, out=None):
        """out[M,N] = (x[M,K] W[N,K]^T + bias) * scale (+ residual)"""
        out = torch.empty(x.shape[0], N, dtype=torch.float32, device=x.device) if out is None else out
        return self._mm(x, W, out, M, N, K, x.stride(0), W.stride(0), bias=bias, scale=scale, residual=residual)

    def _mm(self, A, W, out, M, N, K, lda, ldw, bias=None, scale=None, residual=None, w_trans=False):
        """out[M,N] = (A[M,K] W[N,K]^T + bias) * scale (+ residual).  The decoder's GEMMs over the B*T token rows (2048 x 768)
        are 96 tiles of 128 x 128 -- a third of the chip -- so K is cut into slices computed by separate workgroups (slabs
        summed in a fixed order, then scale / residual: wipa_sum_slabs_ex) whenever the tile grid alone leaves CUs idle."""
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        # two workgroups of this kernel fit a CU: aim at (just under) 512 of them, at least four K-steps each.  Measured on the
        # fine-tune step: 480 workgroups (5 slices of a 96-tile GEMM) 144.9 ms, 288 (3 slices) 147.1, 768 (8 slices) 148.1
        slices = max(1, min(16, 512 // max(tiles, 1), K // 128)) if M >= 512 else 1
        plain = out.dim() == 2 and out.shape[0] == M and out.shape[1] == N and out.is_contiguous() and (
            residual is None or (residual.is_contiguous() and residual.shape == out.shape))
        if slices > 1 and plain:
            need = slices * M * N
            if self._mm_slabs is None or self._mm_slabs.numel() < need:
                self._mm_slabs = torch.empty(need, dtype=torch.float32, device=out.device)
            self._gemm(A, W, self._mm_slabs, M=M, N=N, K=K, lda=lda, ldw=ldw, ldc=N, bias=bias, k_slices=slices, slab_stride=M * N,
                       w_trans=w_trans)
            with on_stream() as s:
                _lib.check(self.L.wipa_sum_slabs_ex(ptr(self._mm_slabs), slices, M * N, ptr(out), M * N, ptr(residual),
                                                    float(scale) if scale is not None else 1.0, sptr(s)), "wipa_sum_slabs_ex")
        else:
            self._gemm(A, W, out, M=M, N=N, K=K, lda=lda, ldw=ldw, ldc=out.stride(0), bias=bias, residual=residual,
                       col_scale_n=(N if scale is not None else 0), col_scale=(scale or 1.0), w_trans=w_trans)
        return out

    def _transpose(self, a, rows, cols, rows_pad):
        """a[rows, cols] (row stride a.stride(0)) -> [cols, rows_pad] with zero padding"""
        out = torch.empty(cols, rows_pad, dtype=a.dtype, device=a.device)
        with on_stream() as s:
            _lib.check(self.L.wipa_transpose(ptr(a), a.stride(0), ptr(out), rows_pad, rows, cols, rows_pad, 0, sptr(s)),
                       "wipa_transpose")
        return out

    def _lin_bwd(self, dy, M, N, x, xT, K, W, dW, db=None, dx=None, accumulate_dx=False, need_dx=True, dy_scaled_ok=True):
        """y = x W^T + b:  dx (+)= dy W ; dW = dy^T x ; db = colsum(dy).
        dy [M(+pad), N], x [M, K]; xT = transpose(x) [K, Mp] may be passed to share it between linears."""
        Mp = _ceil(M, 32)
        out_dx = None
        # wipa_gemm reads K-major operands (a_trans / w_trans): W as it is stored is the K-major form of W^T, dy and x as they
        # are stored are the K-major forms of dy^T and x^T -- no transposed copies.  Needs whole 32-float K-steps and row
        # counts that are multiples of 4; anything else takes the transposing path.
        kmajor = M % 32 == 0 and N % 32 == 0 and K % 4 == 0 and W.stride(1) == 1 and dy.stride(1) == 1 and x.stride(1) == 1
        if need_dx:
            out_dx = dx if dx is not None else torch.empty(dy.shape[0], K, dtype=torch.float32, device=dy.device)
            if kmajor:
                self._mm(dy, W, out_dx, M, K, N, dy.stride(0), W.stride(0), residual=(out_dx if accumulate_dx else None), w_trans=True)
            else:
                Np = _ceil(N, 32)
                WT = self._transpose(W, N, K, Np)  # [K, Np]
                # contraction over n: A = dy (row stride ld, first Np columns must be readable and zero beyond N)
                self._mm(dy, WT, out_dx, M, K, Np, dy.stride(0), Np, residual=(out_dx if accumulate_dx else None))
        if kmajor:
            A_op, W_op, lda, ldw, Kc, flags = dy, x, dy.stride(0), x.stride(0), M, dict(a_trans=True, w_trans=True)
        else:
            dyT = self._transpose(dy, M, N, Mp)  # [N, Mp]
            if xT is None:
                xT = self._transpose(x, M, K, Mp)
            A_op, W_op, lda, ldw, Kc, flags = dyT, xT, Mp, xT.stride(0), Mp, {}
        # weight gradient: small [N, K] output, contraction over all M tokens.  When the tile grid alone cannot fill the
        # chip (768 x 768 over 48 000 encoder positions is 36 tiles), K is split into slabs that are summed in order.
        tiles = ((N + 127) // 128) * ((K + 127) // 128)
        slices = max(1, min(16, 512 // tiles, (Mp // 32) // 8))
        if slices > 1 and dW.is_contiguous():
            need = slices * N * K
            if self._dw_slabs is None or self._dw_slabs.numel() < need:
                self._dw_slabs = torch.empty(need, dtype=torch.float32, device=dy.device)
            self._gemm(A_op, W_op, self._dw_slabs, M=N, N=K, K=Kc, lda=lda, ldw=ldw, ldc=K, k_slices=slices, slab_stride=N * K, **flags)
            with on_stream() as s:
                _lib.check(self.L.wipa_sum_slabs(ptr(self._dw_slabs), slices, N * K, ptr(dW), N * K, 0, sptr(s)), "wipa_sum_slabs")
        else:
            self._gemm(A_op, W_op, dW, M=N, N=K, K=Kc, lda=lda, ldw=ldw, ldc=dW.stride(0), **flags)
        if db is not None:
            self._colsum(dy, M, N, db)
        return out_dx

    def _kv_pair(self, flat: torch.Tensor, pre: str) -> torch.Tensor:
        """cross_attn.key.weight | cross_attn.value.weight of block ``pre`` as one [2d, d] matrix of a flat buffer"""
        d = self.model.dims.n_text_state
        o = self.offsets[f"{pre}.cross_attn.key.weight"]
        assert self.offsets[f"{pre}.cross_attn.value.weight"] == o + d * d
        return flat[o:o + 2 * d * d].view(2 * d, d)

    def _colsum(self, dy, M, N, db):
        with on_stream() as s:
            if self._colsum_ws is None:
                self._colsum_ws = torch.empty(64 * 4 * self.model.dims.n_text_state, dtype=torch.float32, device=dy.device)
            _lib.check(self.L.wipa_colsum(ptr(dy), dy.stride(0), M, N, ptr(db), 0, ptr(self._colsum_ws), self._colsum_ws.numel(),
                                          sptr(s)), "wipa_colsum")

    def _ln(self, x, w, b):
        return ops.layernorm(x, w, b, out_dtype=torch.float32)

    def _ln_bwd(self, x, dy, w, dx, accumulate, dw, db, M, D):
        with on_stream() as s:
            stats = torch.empty(2 * M + 64 * D, dtype=torch.float32, device=x.device)  # (mean, rstd) + chunked dw / db partials
            _lib.check(self.L.wipa_layernorm_bwd(ptr(x), ptr(dy), ptr(w), ptr(dx), int(accumulate), ptr(dw), ptr(db), ptr(stats),
                                                 stats.numel(), M, D, 1e-5, sptr(s)), "wipa_layernorm_bwd")

    def _attn(self, q, k, v, B, H, Tq, Tk, causal, k_rows_per_batch):
        """q [B*Tq, d], k/v [B*Tk, d] row-major -> (out [B*Tq, d], lse [B,H,Tq], desc)"""
        d = H * 64
        with on_stream() as s:
            out = torch.empty(B * Tq, d, dtype=torch.float32, device=q.device)
            lse = torch.empty(B, H, Tq, dtype=torch.float32, device=q.device)
            a = _lib.AttnDesc()
            a.q, a.k, a.v, a.out, a.lse = ptr(q), ptr(k), ptr(v), ptr(out), ptr(lse)
            a.q_bs, a.q_rs, a.q_hs = Tq * q.stride(0), q.stride(0), 64
            a.k_bs, a.k_rs, a.k_hs = Tk * k.stride(0), k.stride(0), 64
            a.v_bs, a.v_rs, a.v_hs = Tk * v.stride(0), v.stride(0), 64
            a.o_bs, a.o_rs, a.o_hs = Tq * d, d, 64
            a.B, a.H, a.Tq, a.Tk, a.causal, a.dtype = B, H, Tq, Tk, int(causal), 0
            _lib.check(self.L.wipa_attention(C.byref(a), sptr(s)), "wipa_attention")
        return out, lse, a

    def _attn_bwd(self, desc, out, d_out, lse, dq, dk, dv):
        with on_stream() as s:
            dvec = torch.empty_like(lse)
            _lib.check(self.L.wipa_attention_bwd(C.byref(desc), ptr(out), ptr(d_out), ptr(lse), ptr(dq), ptr(dk), ptr(dv),
                                                 ptr(dvec), QK_SCALE, sptr(s)), "wipa_attention_bwd")

    # ---- forward + backward ------------------------------------------------------------
    def _forward(self, audio_features: torch.Tensor, tok_in: torch.Tensor):
        """Teacher-forced decoder on ``tok_in`` [B, T] int32 (device, contiguous) and features [B, 1500, d]: returns the
        zero-padded logits [ceil32(B*T), ceil32(V)] f32 and the saved activations the backward needs."""
        m, dm, L = self.model, self.model.dims, self.L
        P = self.p
        B, T = tok_in.shape
        d, H, V, Ta = dm.n_text_state, dm.n_text_head, dm.n_vocab, dm.n_audio_ctx
        M, Mp = B * T, _ceil(B * T, 32)
        Vp = _ceil(V, 32)
        with on_stream() as s:
            dev = m.device
            feats = audio_features.to(device=dev, dtype=torch.float32).contiguous().view(B * Ta, d)
            # [d, ceil32(B*Ta)], shared by all layers; not needed when the weight-gradient GEMM reads feats K-major
            featsT = None if (B * Ta) % 32 == 0 else self._transpose(feats, B * Ta, d, _ceil(B * Ta, 32))
            x = torch.empty(M, d, dtype=torch.float32, device=dev)
            _lib.check(L.wipa_embed_tokens(ptr(tok_in), T, B, T, 0, None, ptr(P("decoder.token_embedding.weight")), 0, None,
                                           ptr(P("decoder.positional_embedding")), ptr(x), d, sptr(s)), "wipa_embed_tokens")
            saved = []
            for l in range(dm.n_text_layer):
                pre = f"decoder.blocks.{l}"
                S = {"x_a": x}
                S["h1"] = self._ln(x, P(f"{pre}.attn_ln.weight"), P(f"{pre}.attn_ln.bias"))
                S["q"] = self._lin(S["h1"], M, d, P(f"{pre}.attn.query.weight"), d, P(f"{pre}.attn.query.bias"), QK_SCALE)
                S["k"] = self._lin(S["h1"], M, d, P(f"{pre}.attn.key.weight"), d, None, QK_SCALE)
                S["v"] = self._lin(S["h1"], M, d, P(f"{pre}.attn.value.weight"), d, P(f"{pre}.attn.value.bias"))
                S["a"], S["lse1"], S["desc1"] = self._attn(S["q"], S["k"], S["v"], B, H, T, T, True, T)
                S["x_b"] = self._lin(S["a"], M, d, P(f"{pre}.attn.out.weight"), d, P(f"{pre}.attn.out.bias"), residual=x)
                S["h2"] = self._ln(S["x_b"], P(f"{pre}.cross_attn_ln.weight"), P(f"{pre}.cross_attn_ln.bias"))
                S["qc"] = self._lin(S["h2"], M, d, P(f"{pre}.cross_attn.query.weight"), d, P(f"{pre}.cross_attn.query.bias"), QK_SCALE)
                # cross key | value as ONE projection over the B*1500 encoder rows: key.weight and value.weight are adjacent in
                # the flat parameter buffer, i.e. one [2d, d] matrix (grid of 750 instead of two of 375 tiles: 2.9 rounds on
                # the 256 CUs instead of 1.5 twice); the key half is scaled in the epilogue, the value half carries its bias
                kv = torch.empty(B * Ta, 2 * d, dtype=torch.float32, device=dev)
                bkv = torch.cat([torch.zeros_like(P(f"{pre}.cross_attn.value.bias")), P(f"{pre}.cross_attn.value.bias")])
                self._gemm(feats, self._kv_pair(self.flat_p, pre), kv, M=B * Ta, N=2 * d, K=d, lda=d, ldw=d, ldc=2 * d, bias=bkv,
                           col_scale_n=d, col_scale=QK_SCALE)
                S["kc"], S["vc"] = kv[:, :d], kv[:, d:]
                S["c"], S["lse2"], S["desc2"] = self._attn(S["qc"], S["kc"], S["vc"], B, H, T, Ta, False, Ta)
                S["x_c"] = self._lin(S["c"], M, d, P(f"{pre}.cross_attn.out.weight"), d, P(f"{pre}.cross_attn.out.bias"),
                                     residual=S["x_b"])
                S["h3"] = self._ln(S["x_c"], P(f"{pre}.mlp_ln.weight"), P(f"{pre}.mlp_ln.bias"))
                S["z"] = self._lin(S["h3"], M, d, P(f"{pre}.mlp1.weight"), 4 * d, P(f"{pre}.mlp1.bias"))
                S["u"] = torch.empty_like(S["z"])
                _lib.check(L.wipa_gelu(ptr(S["z"]), ptr(S["u"]), S["z"].numel(), sptr(s)), "wipa_gelu")
                x = self._lin(S["u"], M, 4 * d, P(f"{pre}.mlp2.weight"), d, P(f"{pre}.mlp2.bias"), residual=S["x_c"])
                saved.append(S)
            x_L = x
            hf = self._ln(x_L, P("decoder.ln.weight"), P("decoder.ln.bias"))
            E = P("decoder.token_embedding.weight")
            logits = torch.zeros(Mp, Vp, dtype=torch.float32, device=dev)  # padding rows/cols stay zero
            self._gemm(hf, E, logits, M=M, N=V, K=d, lda=d, ldw=d, ldc=Vp)
        ctx = dict(saved=saved, x_L=x_L, hf=hf, feats=feats, featsT=featsT, tok_in=tok_in, B=B, T=T)
        return logits, ctx

    def _backward(self, ctx: Dict, dlogits: torch.Tensor, group=None) -> None:
        """Reverse pass from ``dlogits`` [ceil32(B*T), ceil32(V)] f32 (zero in the padding; consumed) into self.flat_g: the
        gradient of every ``decoder.*`` tensor, tied embedding included.  Under DP every finished segment of the flat
        buffer is all-reduced (SUM) asynchronously while the earlier blocks are still in their backward."""
        m, dm, L = self.model, self.model.dims, self.L
        P, G = self.p, self.g
        saved, x_L, hf, feats, featsT, tok_in = ctx["saved"], ctx["x_L"], ctx["hf"], ctx["feats"], ctx["featsT"], ctx["tok_in"]
        B, T = ctx["B"], ctx["T"]
        d, H, V, Ta = dm.n_text_state, dm.n_text_head, dm.n_vocab, dm.n_audio_ctx
        M, Mp = B * T, _ceil(B * T, 32)
        Vp = _ceil(V, 32)
        with on_stream() as s:
            dev = m.device
            self.flat_g.zero_()
            E = P("decoder.token_embedding.weight")
            # logits = hf E^T : dhf = dlogits E ; dE = dlogits^T hf
            ET = self._transpose(E, V, d, Vp)  # [d, Vp]
            dhf = torch.empty(M, d, dtype=torch.float32, device=dev)
            self._mm(dlogits, ET, dhf, M, d, Vp, Vp, Vp)  # 96 tiles over a 51 872-long contraction: eight K slices
            dlT = self._transpose(dlogits, M, V, Mp)  # [V, Mp]
            hfT = self._transpose(hf, M, d, Mp)
            self._gemm(dlT, hfT, G("decoder.token_embedding.weight"), M=V, N=d, K=Mp, lda=Mp, ldw=Mp, ldc=d)
            del dlT, dlogits
            dx = torch.empty(M, d, dtype=torch.float32, device=dev)
            self._ln_bwd(x_L, dhf, P("decoder.ln.weight"), dx, False, G("decoder.ln.weight"), G("decoder.ln.bias"), M, d)
            # DP: each finished gradient segment is all-reduced (SUM; every rank already divided by the GLOBAL
            # count) asynchronously on RCCL's stream while the earlier blocks are still in their backward
            reducer = parallel.SegmentReducer(self.flat_g, group)

            def reduce_segment(rng):
                reducer.reduce(rng[0], rng[1])

            reduce_segment(self.tail_range)
            for l in reversed(range(dm.n_text_layer)):
                pre = f"decoder.blocks.{l}"
                S = saved[l]
                # x_next = x_c + u W2^T + b2
                du = self._lin_bwd(dx, M, d, S["u"], None, 4 * d, P(f"{pre}.mlp2.weight"), G(f"{pre}.mlp2.weight"), G(f"{pre}.mlp2.bias"))
                dz = torch.empty_like(du)
                _lib.check(L.wipa_gelu_bwd(ptr(S["z"]), ptr(du), ptr(dz), dz.numel(), sptr(s)), "wipa_gelu_bwd")
                dh3 = self._lin_bwd(dz, M, 4 * d, S["h3"], None, d, P(f"{pre}.mlp1.weight"), G(f"{pre}.mlp1.weight"), G(f"{pre}.mlp1.bias"))
                self._ln_bwd(S["x_c"], dh3, P(f"{pre}.mlp_ln.weight"), dx, True, G(f"{pre}.mlp_ln.weight"), G(f"{pre}.mlp_ln.bias"), M, d)
                # x_c = x_b + c Wco^T + bco
                dc = self._lin_bwd(dx, M, d, S["c"], None, d, P(f"{pre}.cross_attn.out.weight"), G(f"{pre}.cross_attn.out.weight"),
                                   G(f"{pre}.cross_attn.out.bias"))
                dqc = torch.empty(M, d, dtype=torch.float32, device=dev)
                dkv = torch.empty(B * Ta, 2 * d, dtype=torch.float32, device=dev)  # d(key) | d(value), strides of the forward's kv
                dkc, dvc = dkv[:, :d], dkv[:, d:]
                self._attn_bwd(S["desc2"], S["c"], dc, S["lse2"], dqc, dkc, dvc)
                # one weight-gradient GEMM for the [2d, d] pair (no input gradient: the encoder is frozen); value bias apart
                self._lin_bwd(dkv, B * Ta, 2 * d, feats, featsT, d, self._kv_pair(self.flat_p, pre), self._kv_pair(self.flat_g, pre),
                              None, need_dx=False)
                self._colsum(dvc, B * Ta, d, G(f"{pre}.cross_attn.value.bias"))
                del dkv, dkc, dvc
                dh2 = self._lin_bwd(dqc, M, d, S["h2"], None, d, P(f"{pre}.cross_attn.query.weight"), G(f"{pre}.cross_attn.query.weight"),
                                    G(f"{pre}.cross_attn.query.bias"))
                self._ln_bwd(S["x_b"], dh2, P(f"{pre}.cross_attn_ln.weight"), dx, True, G(f"{pre}.cross_attn_ln.weight"),
                             G(f"{pre}.cross_attn_ln.bias"), M, d)
                # x_b = x_a + a Wo^T + bo
                da = self._lin_bwd(dx, M, d, S["a"], None, d, P(f"{pre}.attn.out.weight"), G(f"{pre}.attn.out.weight"), G(f"{pre}.attn.out.bias"))
                dq = torch.empty(M, d, dtype=torch.float32, device=dev)
                dk = torch.empty(M, d, dtype=torch.float32, device=dev)
                dv = torch.empty(M, d, dtype=torch.float32, device=dev)
                self._attn_bwd(S["desc1"], S["a"], da, S["lse1"], dq, dk, dv)
                h1T = None if M % 32 == 0 else self._transpose(S["h1"], M, d, Mp)  # shared by the three projections
                dh1 = self._lin_bwd(dq, M, d, S["h1"], h1T, d, P(f"{pre}.attn.query.weight"), G(f"{pre}.attn.query.weight"),
                                    G(f"{pre}.attn.query.bias"))
                self._lin_bwd(dk, M, d, S["h1"], h1T, d, P(f"{pre}.attn.key.weight"), G(f"{pre}.attn.key.weight"), None, dx=dh1,
                              accumulate_dx=True)
                self._lin_bwd(dv, M, d, S["h1"], h1T, d, P(f"{pre}.attn.value.weight"), G(f"{pre}.attn.value.weight"),
                              G(f"{pre}.attn.value.bias"), dx=dh1, accumulate_dx=True)
                self._ln_bwd(S["x_a"], dh1, P(f"{pre}.attn_ln.weight"), dx, True, G(f"{pre}.attn_ln.weight"), G(f"{pre}.attn_ln.bias"), M, d)
                saved[l] = None
                reduce_segment(self.block_ranges[l])
            _lib.check(L.wipa_embed_bwd(ptr(tok_in), ptr(dx), B, T, d, ptr(G("decoder.token_embedding.weight")),
                                        ptr(G("decoder.positional_embedding")), sptr(s)), "wipa_embed_bwd")
            if T < dm.n_text_ctx:
                G("decoder.positional_embedding")[T:].zero_()
            reduce_segment(self.head_range)
            # the time the compute stream stalls here is the all-reduce time the backward did NOT hide
            self._ar_events = None
            if reducer.pending:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                reducer.wait()
                e1.record(s)
                self._ar_events = (e0, e1)

    def loss_and_grads(self, audio_features: torch.Tensor, tokens: torch.Tensor, eot: int, group=None):
        """features [B, 1500, d] f32 (encoder output, no gradient), tokens [B, T+1] int.
        Fills self.flat_g (un-clipped, already divided by the global valid count and, under DP,
        all-reduced).  Returns (loss, sum_ce, n_valid) as device scalars.  The fused path of the fine-tune step: masked CE and
        its gradient run in place on the logits (wipa_masked_ce / _bwd), nothing of size [B*T, V] is kept twice."""
        m, dm, L = self.model, self.model.dims, self.L
        B, T1 = tokens.shape
        T = T1 - 1
        V = dm.n_vocab
        M, Vp = B * T, _ceil(V, 32)
        with on_stream() as s:
            dev = m.device
            tok = tokens.to(device=dev, dtype=torch.int32).contiguous()
            tok_in = tok[:, :-1].contiguous()
            logits, ctx = self._forward(audio_features, tok_in)
            row_buf = torch.empty(2 * M, dtype=torch.float32, device=dev)
            stats = torch.empty(2, dtype=torch.float32, device=dev)
            _lib.check(L.wipa_masked_ce(ptr(logits), Vp, ptr(tok), T1, B, T, V, eot, ptr(row_buf), ptr(stats), sptr(s)),
                       "wipa_masked_ce")
            sum_ce, n_valid = parallel.allreduce_loss_stats(stats[0], stats[1], group)
            count = n_valid.reshape(1).contiguous()
            loss = sum_ce / torch.clamp(n_valid, min=1.0)
            _lib.check(L.wipa_masked_ce_bwd(ptr(logits), Vp, ptr(tok), T1, B, T, V, ptr(row_buf[M:]), ptr(count), sptr(s)),
                       "wipa_masked_ce_bwd")
            self._backward(ctx, logits, group)
        return loss, sum_ce, n_valid

    # ---- torch.autograd surface: Whisper.logits differentiable w.r.t. the decoder tensors (train_whisper_ipa.py:232,284) ----
    def leaves(self) -> Dict[str, torch.Tensor]:
        """name -> leaf tensor (requires_grad) that VIEWS the flat parameter buffer: what ``Whisper.logits`` differentiates
        against under torch.enable_grad().  ``.grad`` is filled by ``loss.backward()``; apply_update() reads self.flat_g, so a
        custom training loop copies / accumulates the leaves' gradients there (value_and_grad in scripts/train_whisper_ipa.py)."""
        if self._leaves is None:
            self._leaves = {n: self.p(n).detach().requires_grad_(True) for n in self.names}
        return self._leaves

    differentiable_scope = False  # True inside value_and_grad(model, loss_fn): Whisper.logits then returns differentiable logits

    def differentiable_logits(self, tokens: torch.Tensor, audio_features: torch.Tensor) -> torch.Tensor:
        """logits [B, T, V] f32 with a grad_fn: the HIP forward of loss_and_grads, its hand-written backward behind
        torch.autograd.Function, so ANY torch-written loss on the logits gets exact decoder gradients.  The backward WRITES
        self.flat_g (zeroed first: gradients accumulated there by loss_and_grads are lost) and returns clones of every decoder
        gradient as the leaves' .grad (one more copy of the decoder's size); it reduces over no process group."""
        from . import parallel

        if parallel.world()[1] != 1:
            raise _lib.WipaError("differentiable_logits is single-process (its backward runs without the data-parallel group); "
                                 "use DecoderTrainer.train_step / loss_and_grads(group=...) under torch.distributed")
        leaves = self.leaves()
        return _DecoderLogits.apply(self, tokens, audio_features, *[leaves[n] for n in self.names])


    @property
    def last_allreduce_exposed_ms(self) -> float:
        """Exposed (not overlapped with the backward) gradient all-reduce time of the last step, from HIP events on the
        compute stream around the final waits; 0 for a single process.  Synchronises."""
        ev = getattr(self, "_ar_events", None)
        if not ev:
            return 0.0
        ev[1].synchronize()
        return float(ev[0].elapsed_time(ev[1]))

    def apply_update(self) -> None:
        """per-tensor clip (of the tensors ``clip_scope`` reaches) + AdamW on the flat buffers (flat_g becomes the gradient
        after the clip; ``self.norms`` the un-clipped L2 norm of every tensor)."""
        with on_stream() as s:
            _lib.check(self.L.wipa_clip_adamw(ptr(self.flat_p), ptr(self.flat_g), ptr(self.flat_m), ptr(self.flat_v), ptr(self.ch_off),
                                              ptr(self.ch_len), ptr(self.ch_seg), ptr(self.seg_first), self.n_chunks, self.n_seg,
                                              ptr(self.partial), ptr(self.coef), ptr(self.norms), ptr(self.seg_clip), self.max_grad_norm, self.lr,
                                              self.b1, self.b2, self.eps, self.wd, sptr(s)), "wipa_clip_adamw")
        self.model._invalidate()  # fused inference tables are rebuilt lazily from the updated weights
        self.step_count += 1

    def train_step(self, mel: Optional[torch.Tensor], tokens: torch.Tensor, eot: int, group=None, clip_keys=None):
        """train_whisper_ipa.py:266-311: encoder forward (frozen), loss + grads, clip, AdamW.
        Returns (loss as a device scalar, dict of clipped gradients).
        ``clip_keys`` (one hashable int per clip, e.g. the dataset index) with an enabled feature cache: ``mel`` holds only the
        clips ``feature_cache.missing(clip_keys)`` marks (or is None when every clip is cached); the others come from HBM."""
        if clip_keys is not None and self.feature_cache is not None:
            feats = self.feature_cache.assemble(clip_keys, mel)
        else:
            feats = self.model.embed_audio(mel)
        loss, _, _ = self.loss_and_grads(feats, tokens, eot, group)
        self.apply_update()
        return loss, self.grads()


class _DecoderLogits(torch.autograd.Function):
    """TextDecoder.__call__ of the fine-tune step as one autograd node: forward = DecoderTrainer._forward (libwipa kernels),
    backward = DecoderTrainer._backward.  Gradients flow to the decoder tensors only (the encoder is frozen,
    train_whisper_ipa.py:187; features and tokens get None)."""

    @staticmethod
    def forward(ctx, trainer, tokens, audio_features, *leaves):
        B, T = tokens.shape
        with on_stream():
            tok_in = tokens.to(device=trainer.model.device, dtype=torch.int32).contiguous()
        logits, saved = trainer._forward(audio_features, tok_in)
        ctx.trainer, ctx.saved_acts, ctx.shape = trainer, saved, (B, T, logits.shape[0], logits.shape[1])
        V = trainer.model.dims.n_vocab
        return logits[: B * T].view(B, T, logits.shape[1])[:, :, :V]

    @staticmethod
    def backward(ctx, dlogits):
        tr = ctx.trainer
        B, T, Mp, Vp = ctx.shape
        V = tr.model.dims.n_vocab
        with on_stream():
            dl = torch.zeros(Mp, Vp, dtype=torch.float32, device=tr.model.device)
            dl[: B * T].view(B, T, Vp)[:, :, :V].copy_(dlogits)
        tr._backward(ctx.saved_acts, dl)
        ctx.saved_acts = None
        with on_stream():
            grads = tuple(tr.g(n).clone() for n in tr.names)
        return (None, None, None) + grads


def _numel(shape) -> int:
    n = 1
    for s in shape:
        n *= s
    return n
