"""``Whisper`` with the call surface of ``mlx_whisper.whisper.Whisper`` that the reference's
scripts rely on: ``.dims``, ``.encoder(mel)`` / ``embed_audio``,
``.logits(tokens, features)``, ``.decode``, ``set_dtype``, ``parameters`` / ``update`` in
mlx_whisper's flat dotted key names (train_whisper_ipa.py:43-57,421; transcribe_single.py:18-33).

All arithmetic runs in libwipa.so; this class owns the weight tensors, packs them into the
tables the C++ runtime expects (fused query|key, key|value ... matrices) and owns workspaces.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, asdict
from typing import Dict, List, Optional

import torch

from . import _lib
from .audio import N_FRAMES, PADDED_FRAMES, padded_mel_rows
from .runtime import device, dt_code, on_stream, ptr, ptr_table, sptr, stream, stream_id


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


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    """Positional table of AudioEncoder (computed on the host once, f32)."""
    log_inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-log_inc * torch.arange(channels // 2, dtype=torch.float32))
    t = torch.arange(length, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([torch.sin(t), torch.cos(t)], dim=1)


def _is_matrix(name: str, t: torch.Tensor) -> bool:
    return t.dim() >= 2 and not name.endswith("positional_embedding")


FP8_MAX = 448.0  # largest finite OCP e4m3fn value


def quantize_fp8_e4m3(W: torch.Tensor):
    """[N, ...] -> (codes uint8 [N, K], scale f32 [N]): OCP e4m3fn codes with one POWER-OF-TWO scale per output row,
    value = code * scale.  The scale is the smallest power of two that brings the row's largest magnitude into the
    format's range, so the dequantised value is exactly representable in bf16 (3 mantissa bits, exponent shifted): the
    fp8 weight-streaming kernel and the bf16 tile kernels on the dequantised copy multiply by the SAME numbers."""
    W2 = W.detach().float().reshape(W.shape[0], -1)
    amax = W2.abs().amax(dim=1).clamp(min=2.0 ** -100)
    scale = torch.exp2(torch.ceil(torch.log2(amax / FP8_MAX)))
    q = (W2 / scale[:, None]).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)  # round to nearest even
    return q.view(torch.uint8).contiguous(), scale.contiguous()


def dequantize_fp8_e4m3(codes: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    return codes.view(torch.float8_e4m3fn).float() * scale[:, None].float()


# matrices the decode step streams once per step: kept as fp8 codes by Whisper.quantize_weights
def _decode_step_matrices(dims) -> List[str]:
    names = ["decoder.token_embedding.weight"]
    for i in range(dims.n_text_layer):
        p = f"decoder.blocks.{i}"
        names += [f"{p}.attn.query.weight", f"{p}.attn.key.weight", f"{p}.attn.value.weight", f"{p}.attn.out.weight",
                  f"{p}.cross_attn.query.weight", f"{p}.cross_attn.out.weight", f"{p}.mlp1.weight", f"{p}.mlp2.weight"]
    return names


def parameter_names(dims: ModelDimensions) -> List[str]:
    names = ["encoder.conv1.weight", "encoder.conv1.bias", "encoder.conv2.weight", "encoder.conv2.bias"]

    def block(p, cross):
        out = []
        for a in ["attn"] + (["cross_attn"] if cross else []):
            out += [f"{p}.{a}.query.weight", f"{p}.{a}.query.bias", f"{p}.{a}.key.weight", f"{p}.{a}.value.weight",
                    f"{p}.{a}.value.bias", f"{p}.{a}.out.weight", f"{p}.{a}.out.bias", f"{p}.{a}_ln.weight", f"{p}.{a}_ln.bias"]
        out += [f"{p}.mlp1.weight", f"{p}.mlp1.bias", f"{p}.mlp2.weight", f"{p}.mlp2.bias", f"{p}.mlp_ln.weight", f"{p}.mlp_ln.bias"]
        return out

    for i in range(dims.n_audio_layer):
        names += block(f"encoder.blocks.{i}", False)
    names += ["encoder.ln_post.weight", "encoder.ln_post.bias", "decoder.token_embedding.weight", "decoder.positional_embedding"]
    for i in range(dims.n_text_layer):
        names += block(f"decoder.blocks.{i}", True)
    names += ["decoder.ln.weight", "decoder.ln.bias"]
    return names


class _Part:
    """``model.encoder`` / ``model.decoder``: callable + freeze()/unfreeze() like mlx.nn.Module."""

    def __init__(self, model: "Whisper", prefix: str):
        self._model, self._prefix = model, prefix

    def freeze(self):
        self._model._frozen.add(self._prefix)
        return self

    def unfreeze(self):
        self._model._frozen.discard(self._prefix)
        return self

    def __call__(self, *args, **kw):
        if self._prefix == "encoder":
            return self._model.embed_audio(*args, **kw)
        return self._model.logits(*args, **kw)


class Whisper:
    def __init__(self, dims: ModelDimensions, dtype: torch.dtype = torch.float32, f32_split: Optional[bool] = None,
                 sinusoid_rounding: str = "f32", cross_attention: str = "auto", cross_splits: int = 0):
        """``f32_split`` (float32 models only): the large GEMMs and the encoder attention take every f32 product as three
        split-bf16 MFMA terms (~5e-6 relative error per dot product) instead of on the f32 MFMA.  ON by default since round 5
        (None = on; False = exact f32 products, what the reference computes: train_whisper_ipa.py:505, transcribe_single.py:13).
        The decision was taken on measurement: whisper-small 12+12 against the float32 CPU checker with split
        products -- features 5.9e-5, logits 2.1e-4, loss 8.3e-6 (north_star: logits / loss within 1e-3), all 64 greedy ids of
        both clips bit-exact, decoder gradients within 2.7e-5 relative; exact products on the same run: 9.6e-6 / 3.0e-5 / 1.9e-6,
        i.e. split spends 21 % of the logit tolerance where exact spends 3 % (tests/test_gpu_full_depth.py, r05).  Every float32
        parity test of the repo passes in both modes (tests/conftest.py f32_mode; the unparametrised ones run the default).
        Cost of exact: 292 against 204 ms per 64-clip pass, 145 against 87 ms per 32-clip fine-tune step.
        ``sinusoid_rounding``: "f32" (default) adds the encoder's sinusoid table as computed in float32, which is what the
        published algorithm does; "fp16" rounds the table to fp16 first -- SURVEY.md App. C.3: mlx_whisper builds
        ``_positional_embedding`` in the load dtype (fp16) and ``set_dtype(float32)`` does not touch the private attribute
        [UPSTREAM-UNVERIFIED], so a reference run may carry the rounded table (features move by ~1e-3).
        ``cross_attention``: how a decode step reads the audio.  "cached": K = xa Wk^T and V = xa Wv^T + bv are projected once
        per decoder layer and streamed every step, as mlx_whisper caches them.  "absorbed": Wk is absorbed into the query and
        Wv into the output, and every layer streams the encoder output xa itself -- half the bytes per step, no K/V cache, no
        projection GEMMs (csrc/cross_absorbed.hip); bf16 models with <= 16 heads and d in {384, 512, 768, 1024}, no fp8 tables.
        "auto" (default): per decode call, from (clips, new tokens) by the measured table (use_absorbed): absorbed where it
        applies, except long outputs at a small batch (<= 64 clips with >= 192 new tokens -- the reference's own default
        sample_len of 224 at the benchmark batch), where cached K / V is faster; WIPA_CROSS_ABSORB=0 turns absorbed off.
        Same mathematics, other bf16 rounding points.  Measured on MI355X, absorbed against cached:
        whisper-small, 64 clips: +5.3 % at 32, +7.0 % at 64, +4.5 % at 128, -3.1 % at 224 new tokens; 128 clips: +8.5 % / +3.6 % at
        64 / 224; whisper-medium, 256 clips: +1.1 % at 224 (profiles/r04_cached_vs_absorbed.txt).  The choice is
        explicit here and in the CPU checker, and both settings are pinned by golden fixtures (tests/golden/wide_model.npz).
        ``cross_splits`` (absorbed form only; also an attribute that may be changed between decodes): frame splits per clip of the
        decode step's streaming launch.  0 = the library default, 4: the shortest launch for a decode that has the GPU to itself
        (one clip: 4 workgroups of 375 frames).  2 = half-chip launches for callers that keep SEVERAL passes in flight on one
        GPU (bench.py's pipelined passes): the streaming kernels of two passes run side by side instead of queueing for all 256
        CUs -- measured on MI355X, whisper-small, 64 clips, 4 passes in flight: 72.3 against 75.3 ms per pass, while a lone
        decode step costs 1.35 against 1.25 ms (3: 73.5 ms / 1.26 ms; profiles/r04_stream_splits_ab.txt).  The count never
        depends on the batch; it moves the order of the softmax merges, i.e. bf16-level rounding like any other kernel choice."""
        if cross_splits not in (0, 1, 2, 3, 4):
            raise _lib.WipaError(f"cross_splits must be 0 (default) or 1..4, got {cross_splits!r}")
        self._cross_splits = int(cross_splits)
        if cross_attention not in ("auto", "absorbed", "cached"):
            raise _lib.WipaError(f"cross_attention must be 'auto', 'absorbed' or 'cached', got {cross_attention!r}")
        self.cross_attention = cross_attention
        if sinusoid_rounding not in ("f32", "fp16"):
            raise _lib.WipaError(f"sinusoid_rounding must be 'f32' or 'fp16', got {sinusoid_rounding!r}")
        self.sinusoid_rounding = sinusoid_rounding
        self.dims = dims
        self.dtype = dtype
        self.f32_split = True if f32_split is None else bool(f32_split)  # only read for float32 (see _cfg)
        self.device = device()
        _lib.lib()  # fail now if the extension is missing
        self._params: Dict[str, torch.Tensor] = {}
        self._fp8: Dict[str, tuple] = {}  # name -> (codes uint8 [N,K], scale f32 [N]) for the decode-step matrices
        self._fp8_enc: Dict[str, tuple] = {}  # ... and for the encoder's q|k, value, mlp1, mlp2 when activations are fp8 too
        self._frozen = set()
        self._enc_generation = 0  # bumped whenever an encoder tensor changes: FrozenFeatureCache entries are functions of it
        self._packed = None
        self._packed_tf = None
        self._packed_abs = None
        self._enc_ws: Dict[int, torch.Tensor] = {}  # per library stream
        self._tf_ws: Dict[int, torch.Tensor] = {}
        self.encoder = _Part(self, "encoder")
        self.decoder = _Part(self, "decoder")
        self.training = False
        with on_stream():
            pos = sinusoids(dims.n_audio_ctx, dims.n_audio_state)
            if sinusoid_rounding == "fp16":
                pos = pos.to(torch.float16).to(torch.float32)
            self._enc_pos = pos.to(self.device)

    # ---- mlx.nn.Module-like surface ------------------------------------------------
    @property
    def is_multilingual(self) -> bool:
        return self.dims.n_vocab >= 51865

    @property
    def num_languages(self) -> int:
        return self.dims.n_vocab - 51765 - int(self.is_multilingual)

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def set_dtype(self, dtype: torch.dtype):
        """Matrices (and activations / KV caches) switch to ``dtype``; biases, LayerNorm and
        positional tables stay f32 (train_whisper_ipa.py:505, transcribe_single.py:13)."""
        dt_code(dtype)
        self.dtype = dtype
        self._fp8 = {}  # fp8 codes belong to the bf16 configuration they were made for (quantize_weights again if wanted)
        self._fp8_enc = {}
        with on_stream():
            for k, v in list(self._params.items()):
                if _is_matrix(k, v):
                    self._params[k] = v.to(dtype)
        self._enc_generation += 1
        self._invalidate()
        return self

    def load_weights(self, flat: Dict[str, torch.Tensor], strict: bool = True):
        """``flat``: mlx_whisper key names (App. A.5 of SURVEY.md).  Conv weights [d,3,c_in]."""
        names = parameter_names(self.dims)
        missing = [n for n in names if n not in flat and n not in self._params]
        if strict and missing:
            raise KeyError(f"missing weights: {missing[:5]} ... ({len(missing)})")
        with on_stream():
            for n in names:
                if n in flat:
                    t = torch.as_tensor(flat[n]).detach()
                    want = self.dtype if _is_matrix(n, t) else torch.float32
                    self._params[n] = t.to(device=self.device, dtype=want).contiguous()
                    if n.startswith("encoder."):
                        self._enc_generation += 1
        self._fp8 = {}  # new weights: any fp8 codes are stale (call quantize_weights again)
        self._fp8_enc = {}
        self._invalidate()
        return self

    # ---- fp8 weights (BASELINE.json configs[4]: whisper-large-v3 fp8-weight inference) -------------
    @property
    def weights_format(self) -> str:
        return "fp8_e4m3" if self._fp8 else str(self.dtype).replace("torch.", "")

    @property
    def activations_format(self) -> str:
        """"fp8_e4m3" when the encoder's q|k, value, mlp1 and mlp2 projections run fp8 x fp8 (quantize_weights(activations="fp8"))"""
        return "fp8_e4m3" if self._fp8_enc else str(self.dtype).replace("torch.", "")

    def quantize_weights(self, fmt: str = "fp8_e4m3", activations: str = "bf16"):
        """``activations="fp8"`` additionally runs the ENCODER's q|k, value, mlp1 and mlp2 projections (11/12 of its GEMM FLOPs)
        with e4m3 activations on the block-scaled fp8 MFMA: LayerNorm outputs and the GELU output are quantised per row
        (power-of-two scales), the weights are the e4m3 codes (wipa_model_cfg.enc_act_fp8).  This is NOT the same model as
        "bf16 activations on quantised weights": expect feature differences of a few percent (tests/test_gpu_model.py states
        the measured bound).  Default "bf16".

        Quantise EVERY matrix of the model (Linear / Conv1d weights, token embedding) to OCP fp8 e4m3fn with one
        power-of-two scale per output row (quantize_fp8_e4m3).  Biases, LayerNorm and positional tables stay f32.
        The matrices the decode step streams once per step stay in fp8 (1 byte per weight) and are read by the fp8
        weight-streaming GEMM; all matrices are ALSO kept as their exact bf16 dequantisation for the MFMA-bound tile GEMMs of
        the encoder, the cross-K/V projection and the teacher-forced decoder -- both forms multiply by the same values."""
        if fmt != "fp8_e4m3":
            raise _lib.WipaError(f"quantize_weights: unknown format {fmt!r} (fp8_e4m3)")
        if self.dtype != torch.bfloat16:
            raise _lib.WipaError("quantize_weights: fp8 weights run with bf16 activations: call set_dtype(torch.bfloat16) first")
        if activations not in ("bf16", "fp8"):
            raise _lib.WipaError(f"quantize_weights: activations must be 'bf16' or 'fp8', got {activations!r}")
        if activations == "fp8" and self.dims.n_audio_state % 128 != 0:
            raise _lib.WipaError("quantize_weights: fp8 activations need an encoder width that is a multiple of 128")
        step = set(_decode_step_matrices(self.dims))
        enc_f8 = set()
        if activations == "fp8":
            for i in range(self.dims.n_audio_layer):
                pfx = f"encoder.blocks.{i}"
                enc_f8 |= {f"{pfx}.attn.query.weight", f"{pfx}.attn.key.weight", f"{pfx}.attn.value.weight", f"{pfx}.mlp1.weight",
                           f"{pfx}.mlp2.weight"}
        fp8, fp8_enc = {}, {}
        with on_stream():
            for n, t in list(self._params.items()):
                if not _is_matrix(n, t):
                    continue
                codes, scale = quantize_fp8_e4m3(t)
                self._params[n] = dequantize_fp8_e4m3(codes, scale).to(torch.bfloat16).reshape(t.shape).contiguous()
                if n in step:
                    fp8[n] = (codes, scale)
                if n in enc_f8:
                    fp8_enc[n] = (codes, scale)
        self._enc_generation += 1
        self._invalidate()
        self._fp8 = fp8
        self._fp8_enc = fp8_enc
        return self

    def parameters(self) -> Dict:
        return _unflatten(self._params)

    def trainable_parameters(self) -> Dict:
        keep = {k: v for k, v in self._params.items() if k.split(".")[0] not in self._frozen}
        return _unflatten(keep)

    def flat_parameters(self) -> Dict[str, torch.Tensor]:
        return dict(self._params)

    def update(self, tree: Dict):
        self.load_weights(_flatten(tree), strict=False)
        return self

    def _invalidate(self):
        """The weight tensors changed (load / update / dtype / optimiser step): drop the packed tables AND every decode-step
        graph captured against them -- a graph holds the device pointers of the fused q|k|v matrices that die with the old
        tables, and the table's host address alone (part of the graph key) can be reused by the next one."""
        if self._packed is not None or self._packed_abs is not None or getattr(self, "_dec_states", None):
            torch.cuda.synchronize(self.device)  # nothing may still be replaying a graph we are about to destroy
        for st in getattr(self, "_dec_states", {}).values():
            st.release()
        self._generation = (getattr(self, "_generation", 0) + 1) & 0x7FFFFFFF
        self._packed = None
        self._packed_tf = None
        self._packed_abs = None

    # "auto" by the measured table (profiles/r04_cached_vs_absorbed.txt, MI355X, 4 passes in flight; absorbed against cached):
    # whisper-small 64 clips: +5.3 % at 32 new tokens, +7.0 % at 64, +4.5 % at 128, -3.1 % at 224; 128 clips: +8.5 % / +3.6 %
    # at 64 / 224; whisper-medium 256 clips: +1.1 % at 224 (r03: +5.8 % at 64).  The one losing regime is long outputs at a small batch (every
    # pass in flight sits in its decode loop and the absorbed step has two more launches per layer), and 224 = n_text_ctx // 2 is
    # what the reference's callers get by default (scripts/transcribe_single.py:49-52 leaves sample_len unset).
    AUTO_CACHED_MAX_BATCH = 64
    AUTO_CACHED_MIN_NEW_TOKENS = 192

    @property
    def cross_absorbed_eligible(self) -> bool:
        d = self.dims
        return (self.dtype == torch.bfloat16 and not self._fp8 and d.n_text_head <= 16 and d.n_text_state in (384, 512, 768, 1024)
                and d.n_text_state == d.n_audio_state)

    def use_absorbed(self, batch: Optional[int] = None, new_tokens: Optional[int] = None) -> bool:
        """whether a decode of ``batch`` clips x ``new_tokens`` positions uses the absorbed-projection cross-attention (see
        ``cross_attention`` in __init__).  "absorbed" / "cached" force the form; "auto" takes absorbed where it applies EXCEPT
        for long outputs at a small batch (<= 64 clips with >= 192 new tokens: the measured table above), where the cached
        K / V form is faster.  Unknown sizes (None) count as short / large.  WIPA_CROSS_ABSORB=0 turns "auto" off."""
        import os

        eligible = self.cross_absorbed_eligible
        if self.cross_attention == "absorbed":
            if not eligible:
                raise _lib.WipaError("cross_attention='absorbed' needs a bf16 model without fp8 tables, <= 16 heads, d in {384, 512, 768, 1024}")
            return True
        if self.cross_attention == "cached" or not eligible or os.environ.get("WIPA_CROSS_ABSORB", "1") == "0":
            return False
        long_small = (batch is not None and new_tokens is not None and batch <= self.AUTO_CACHED_MAX_BATCH
                      and new_tokens >= self.AUTO_CACHED_MIN_NEW_TOKENS)
        return not long_small

    @property
    def cross_splits(self) -> int:
        return self._cross_splits

    @cross_splits.setter
    def cross_splits(self, n: int):
        if n not in (0, 1, 2, 3, 4):
            raise _lib.WipaError(f"cross_splits must be 0 (default) or 1..4, got {n!r}")
        self._cross_splits = int(n)
        if self._packed_abs is not None:
            self._packed_abs["cfg"].dec_cross_splits = self._cross_splits  # part of the step graph's key: the next decode re-captures

    @property
    def cross_absorbed(self) -> bool:
        """the form a decode of unknown size takes (use_absorbed()); decoding.py asks use_absorbed(B, sample_len) per call"""
        return self.use_absorbed()

    # ---- packing -------------------------------------------------------------------
    def _cfg(self, fp8: bool = False, absorbed: bool = False) -> _lib.ModelCfg:
        d = self.dims
        return _lib.ModelCfg(d.n_mels, d.n_audio_ctx, d.n_audio_state, d.n_audio_head, d.n_audio_layer, d.n_vocab,
                             d.n_text_ctx, d.n_text_state, d.n_text_head, d.n_text_layer, dt_code(self.dtype),
                             int(self.f32_split and self.dtype == torch.float32), _lib.WIPA_FP8_E4M3 if fp8 else 0,
                             getattr(self, "_generation", 0), int(bool(self._fp8_enc)), int(bool(absorbed) and not fp8),
                             self._cross_splits if (absorbed and not fp8) else 0)

    def packed(self, teacher_forced: bool = False, absorbed: Optional[bool] = None):
        """(cfg, encoder table, decoder table); fused matrices are rebuilt after any update.  With fp8 weights the decoder
        table carries the e4m3 codes + scales of the decode-step matrices; ``teacher_forced=True`` gives the all-bf16 table
        (dequantised values) that wipa_decoder_logits needs.  ``absorbed`` (default: use_absorbed() of an unknown size) selects
        the decode-step cross-attention form: both variants SHARE every weight tensor -- the absorbed one appends Wk^T per
        layer to the decoder table and sets cfg.dec_cross_absorbed -- and both stay alive until the next weight change, so a
        captured step graph never outlives the table it points into."""
        fp8 = bool(self._fp8) and not teacher_forced
        if teacher_forced and self._fp8:
            if self._packed_tf is None:
                self._packed_tf = self._pack(False)
            return self._packed_tf
        if self._packed is None:
            self._packed = self._pack(fp8)
        if absorbed is None:
            absorbed = self.use_absorbed()
        if not absorbed or fp8:
            return self._packed
        if not self.cross_absorbed_eligible:
            raise _lib.WipaError("packed(absorbed=True): the absorbed cross-attention does not apply to this model")
        if self._packed_abs is None:
            base, d = self._packed, self.dims
            with on_stream():
                wkT = [self._params[f"decoder.blocks.{i}.cross_attn.key.weight"].to(self.dtype).t().contiguous()
                       for i in range(d.n_text_layer)]  # WIPA_DEC_ABSORBED_PER_LAYER: Wk^T per layer after the regular blocks
            stream().synchronize()
            dec = list(base["dec"]) + wkT
            self._packed_abs = dict(cfg=self._cfg(False, True), enc=base["enc"], dec=dec, enc_tab=base["enc_tab"], dec_tab=ptr_table(dec))
        return self._packed_abs

    def _pack(self, fp8: bool):
        P, T, d = self._params, self.dtype, self.dims
        kmul = 64 if T == torch.bfloat16 else 32
        f32 = torch.float32
        with on_stream():
            def mat(x):
                return x.to(T).contiguous()

            def vec(x):
                return x.to(f32).contiguous()

            de = d.n_audio_state
            K1 = (3 * d.n_mels + kmul - 1) // kmul * kmul
            c1 = torch.zeros(de, K1, dtype=T, device=self.device)
            c1[:, : 3 * d.n_mels] = P["encoder.conv1.weight"].reshape(de, 3 * d.n_mels).to(T)
            enc = [c1, vec(P["encoder.conv1.bias"]), mat(P["encoder.conv2.weight"].reshape(de, 3 * de)),
                   vec(P["encoder.conv2.bias"]), vec(self._enc_pos), vec(P["encoder.ln_post.weight"]),
                   vec(P["encoder.ln_post.bias"])]
            for i in range(d.n_audio_layer):
                p = f"encoder.blocks.{i}"
                qb = P[f"{p}.attn.query.bias"]
                enc += [vec(P[f"{p}.attn_ln.weight"]), vec(P[f"{p}.attn_ln.bias"]),
                        mat(torch.cat([P[f"{p}.attn.query.weight"], P[f"{p}.attn.key.weight"]], 0)),
                        vec(torch.cat([qb.float(), torch.zeros_like(qb, dtype=f32)], 0)),
                        mat(P[f"{p}.attn.value.weight"]), vec(P[f"{p}.attn.value.bias"]),
                        mat(P[f"{p}.attn.out.weight"]), vec(P[f"{p}.attn.out.bias"]),
                        vec(P[f"{p}.mlp_ln.weight"]), vec(P[f"{p}.mlp_ln.bias"]),
                        mat(P[f"{p}.mlp1.weight"]), vec(P[f"{p}.mlp1.bias"]),
                        mat(P[f"{p}.mlp2.weight"]), vec(P[f"{p}.mlp2.bias"])]
            dec = [mat(P["decoder.token_embedding.weight"]), vec(P["decoder.positional_embedding"]),
                   vec(P["decoder.ln.weight"]), vec(P["decoder.ln.bias"])]
            for i in range(d.n_text_layer):
                p = f"decoder.blocks.{i}"
                qb, vb = P[f"{p}.attn.query.bias"].float(), P[f"{p}.attn.value.bias"].float()
                cvb = P[f"{p}.cross_attn.value.bias"].float()
                z = torch.zeros_like(qb)
                dec += [vec(P[f"{p}.attn_ln.weight"]), vec(P[f"{p}.attn_ln.bias"]),
                        mat(torch.cat([P[f"{p}.attn.query.weight"], P[f"{p}.attn.key.weight"], P[f"{p}.attn.value.weight"]], 0)),
                        vec(torch.cat([qb, z, vb], 0)),
                        mat(P[f"{p}.attn.out.weight"]), vec(P[f"{p}.attn.out.bias"]),
                        vec(P[f"{p}.cross_attn_ln.weight"]), vec(P[f"{p}.cross_attn_ln.bias"]),
                        mat(P[f"{p}.cross_attn.query.weight"]), vec(P[f"{p}.cross_attn.query.bias"]),
                        mat(torch.cat([P[f"{p}.cross_attn.key.weight"], P[f"{p}.cross_attn.value.weight"]], 0)),
                        vec(torch.cat([z, cvb], 0)),
                        mat(P[f"{p}.cross_attn.out.weight"]), vec(P[f"{p}.cross_attn.out.bias"]),
                        vec(P[f"{p}.mlp_ln.weight"]), vec(P[f"{p}.mlp_ln.bias"]),
                        mat(P[f"{p}.mlp1.weight"]), vec(P[f"{p}.mlp1.bias"]),
                        mat(P[f"{p}.mlp2.weight"]), vec(P[f"{p}.mlp2.bias"])]
            assert len(enc) == _lib.ENC_GLOBAL + _lib.ENC_PER_LAYER * d.n_audio_layer
            if self._fp8_enc:  # fp8 encoder tail: codes + per-row scales of q|k, value, mlp1, mlp2 (WIPA_ENC_FP8_PER_LAYER)
                E = self._fp8_enc
                for i in range(d.n_audio_layer):
                    p = f"encoder.blocks.{i}"
                    for names in ((f"{p}.attn.query.weight", f"{p}.attn.key.weight"), (f"{p}.attn.value.weight",), (f"{p}.mlp1.weight",),
                                  (f"{p}.mlp2.weight",)):
                        enc.append(torch.cat([E[n][0] for n in names], 0).to(self.device).contiguous())
                        enc.append(torch.cat([E[n][1] for n in names], 0).to(device=self.device, dtype=f32).contiguous())
                assert len(enc) == _lib.ENC_GLOBAL + (_lib.ENC_PER_LAYER + _lib.ENC_FP8_PER_LAYER) * d.n_audio_layer
            assert len(dec) == _lib.DEC_GLOBAL + _lib.DEC_PER_LAYER * d.n_text_layer
            if fp8:
                # the decode-step matrices as e4m3 codes (rows concatenated like their bf16 counterparts) + per-row scales
                F = self._fp8

                def codes(*names):
                    return torch.cat([F[n][0] for n in names], 0).to(self.device).contiguous()

                def scales(*names):
                    return torch.cat([F[n][1] for n in names], 0).to(device=self.device, dtype=f32).contiguous()

                dec[0] = codes("decoder.token_embedding.weight")
                tail = [scales("decoder.token_embedding.weight")]
                for i in range(d.n_text_layer):
                    p = f"decoder.blocks.{i}"
                    groups = [(f"{p}.attn.query.weight", f"{p}.attn.key.weight", f"{p}.attn.value.weight"), (f"{p}.attn.out.weight",),
                              (f"{p}.cross_attn.query.weight",), (f"{p}.cross_attn.out.weight",), (f"{p}.mlp1.weight",),
                              (f"{p}.mlp2.weight",)]
                    for slot, g in zip((2, 4, 8, 12, 16, 18), groups):
                        dec[_lib.DEC_GLOBAL + _lib.DEC_PER_LAYER * i + slot] = codes(*g)
                        tail.append(scales(*g))
                dec += tail
                assert len(tail) == 1 + _lib.DEC_FP8_PER_LAYER * d.n_text_layer
        stream().synchronize()  # the tables are read from every library stream
        return dict(cfg=self._cfg(fp8, False), enc=enc, dec=dec, enc_tab=ptr_table(enc), dec_tab=ptr_table(dec))

    # ---- encoder -------------------------------------------------------------------
    def encode_padded(self, mel_padded: torch.Tensor, B: int) -> torch.Tensor:
        """padded mel [B*3002+4, n_mels] in the model dtype -> features [B, 1500, d]."""
        L = _lib.lib()
        pk = self.packed()
        assert mel_padded.dtype == self.dtype and mel_padded.shape[0] >= padded_mel_rows(B)
        with on_stream() as s:
            need = L.wipa_encoder_workspace_bytes(C.byref(pk["cfg"]), B)
            ws = self._enc_ws.get(stream_id())
            if ws is None or ws.numel() < need:
                ws = torch.empty(need, dtype=torch.uint8, device=self.device)
                self._enc_ws[stream_id()] = ws
            out = torch.empty(B, self.dims.n_audio_ctx, self.dims.n_audio_state, dtype=self.dtype, device=self.device)
            _lib.check(L.wipa_encoder_forward(C.byref(pk["cfg"]), pk["enc_tab"], ptr(mel_padded), ptr(out), ptr(ws), ws.numel(),
                                              B, sptr(s)), "wipa_encoder_forward")
        return out

    def embed_audio(self, mel: torch.Tensor) -> torch.Tensor:
        """mel [B, 3000, n_mels] (or [3000, n_mels]) -> [B, 1500, d]  (train_whisper_ipa.py:223)."""
        L = _lib.lib()
        if mel.dim() == 2:
            mel = mel[None]
        B = mel.shape[0]
        assert mel.shape[1] == N_FRAMES and mel.shape[2] == self.dims.n_mels, mel.shape
        with on_stream() as s:
            mel32 = mel.to(device=self.device, dtype=torch.float32).contiguous()
            padded = torch.empty(padded_mel_rows(B), self.dims.n_mels, dtype=self.dtype, device=self.device)
            _lib.check(L.wipa_mel_pad_cast(ptr(mel32), B, self.dims.n_mels, ptr(padded), dt_code(self.dtype), sptr(s)),
                       "wipa_mel_pad_cast")
        return self.encode_padded(padded, B)

    # ---- teacher-forced decoder ----------------------------------------------------
    def logits(self, tokens: torch.Tensor, audio_features: torch.Tensor, differentiable: Optional[bool] = None) -> torch.Tensor:
        """tokens [B,T] int, features [B,1500,d] -> logits [B,T,V] f32 (train_whisper_ipa.py:232).

        By default the forward-only C++ pass over the packed tables -- also in torch's (default-on) grad mode: an evaluation
        caller outside ``torch.no_grad()`` gets no saved-activation forward and no grad_fn behind its back.
        DIFFERENTIABLE w.r.t. the decoder tensors in exactly two cases, both needing a float32 model whose
        decoder parameters a ``DecoderTrainer`` owns: (1) inside ``value_and_grad(model, loss_fn)`` of
        scripts/train_whisper_ipa.py -- the reference's ``nn.value_and_grad`` (:284) is what makes ITS model call
        differentiable, so a loss function written like the reference's needs no change; (2) ``differentiable=True``.  The call
        then goes through ``training._DecoderLogits`` (HIP forward with saved activations, hand-written HIP backward) and
        ``loss.backward()`` on ANY torch-written loss fills the ``.grad`` of ``trainer.leaves()``.  NOTE: that backward
        OVERWRITES the trainer's flat gradient buffer (it does not accumulate onto gradients left there by
        ``loss_and_grads``) and runs without a data-parallel group: single-process use; the DP step is ``train_step``."""
        tr = getattr(self, "_trainer", None)
        if differentiable is None:
            differentiable = tr is not None and tr.differentiable_scope and torch.is_grad_enabled()
        if differentiable:
            if tr is None or self.dtype != torch.float32:
                raise _lib.WipaError("Whisper.logits(differentiable=True) needs a float32 model and a DecoderTrainer(model) owning its decoder")
            return tr.differentiable_logits(tokens, audio_features)
        L = _lib.lib()
        pk = self.packed(teacher_forced=True)
        B, T = tokens.shape
        V = self.dims.n_vocab
        ldl = (V + 7) // 8 * 8
        with on_stream() as s:
            tok = tokens.to(device=self.device, dtype=torch.int32).contiguous()
            feats = audio_features.to(device=self.device, dtype=self.dtype).contiguous()
            need = L.wipa_decoder_logits_workspace_bytes(C.byref(pk["cfg"]), B, T)
            tf_ws = self._tf_ws.get(stream_id())
            if tf_ws is None or tf_ws.numel() < need:
                tf_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
                self._tf_ws[stream_id()] = tf_ws
            out = torch.empty(B * T, ldl, dtype=torch.float32, device=self.device)
            _lib.check(L.wipa_decoder_logits(C.byref(pk["cfg"]), pk["dec_tab"], ptr(tok), ptr(feats), ptr(out), ldl,
                                             ptr(tf_ws), tf_ws.numel(), B, T, sptr(s)), "wipa_decoder_logits")
        return out.view(B, T, ldl)[:, :, :V]

    def __call__(self, mel: torch.Tensor, tokens: torch.Tensor) -> torch.Tensor:
        return self.logits(tokens, self.embed_audio(mel))

    def decode(self, mel_or_features: torch.Tensor, options=None):
        from .decoding import DecodingOptions, decode

        return decode(self, mel_or_features, options or DecodingOptions())


def _flatten(tree, prefix=""):
    out = {}
    if isinstance(tree, dict):
        for k, v in tree.items():
            out.update(_flatten(v, f"{prefix}{k}."))
    elif isinstance(tree, (list, tuple)):
        for i, v in enumerate(tree):
            out.update(_flatten(v, f"{prefix}{i}."))
    else:
        out[prefix[:-1]] = tree
    return out


def _unflatten(flat: Dict[str, torch.Tensor]) -> Dict:
    root: Dict = {}
    for key, v in flat.items():
        parts = key.split(".")
        node = root
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v

    def listify(node):
        if isinstance(node, dict):
            node = {k: listify(v) for k, v in node.items()}
            if node and all(k.isdigit() for k in node):
                return [node[str(i)] for i in range(len(node))]
        return node

    return listify(root)
