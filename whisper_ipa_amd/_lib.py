"""ctypes binding of libwipa.so (C ABI in include/wipa.h).

There is NO fallback: if the shared library is missing or a symbol is absent the
import of any compute entry point raises.  Build it with ``python -c "import
__graft_entry__ as g; g.build()"`` from the repo root.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwipa.so")

WIPA_F32, WIPA_BF16, WIPA_FP8_E4M3 = 0, 1, 2
GREEDY_PARTS = 2048  # WIPA_GREEDY_PARTS
ENC_GLOBAL, ENC_PER_LAYER = 7, 14
DEC_GLOBAL, DEC_PER_LAYER, DEC_FP8_PER_LAYER = 4, 20, 6
GEMM_DISPATCH = ("tile128", "tile256", "tile384", "tile384n", "tile256p", "skinny", "skinny_fp8", "skinny_ln", "kmajor", "split_k",
                 "tile_fp8", "tile_fp8_384")
ENC_FP8_PER_LAYER = 8
DEC_ABSORBED_PER_LAYER = 1

c_void_p, c_int, c_int64, c_size_t, c_float = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", c_void_p), ("W", c_void_p), ("C", c_void_p), ("bias", c_void_p), ("residual", c_void_p),
        ("pos", c_void_p), ("c_offset_dev", c_void_p),
        ("lda", c_int64), ("ldw", c_int64), ("ldc", c_int64), ("ldpos", c_int64),
        ("rg_stride", c_int64), ("cg_stride", c_int64), ("c_offset", c_int64), ("slab_stride", c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("in_dtype", C.c_int32), ("out_dtype", C.c_int32),
        ("rg_in", C.c_int32), ("rg_valid", C.c_int32), ("cg_in", C.c_int32),
        ("zero_invalid_rows", C.c_int32), ("bias_along_m", C.c_int32), ("act", C.c_int32),
        ("col_scale_n", C.c_int32), ("col_scale", c_float), ("k_slices", C.c_int32),
        ("f32_split", C.c_int32),
        ("w_scale", c_void_p), ("w_dtype", C.c_int32),
        ("ln_x", c_void_p), ("ln_w", c_void_p), ("ln_b", c_void_p), ("ln_ldx", c_int64), ("ln_eps", c_float),
        ("stream_weights", C.c_int32), ("a_trans", C.c_int32), ("w_trans", C.c_int32),
        ("a_scale", c_void_p),
    ]


class SelfBlockDesc(C.Structure):
    _fields_ = [
        ("x", c_void_p), ("ln_w", c_void_p), ("ln_b", c_void_p), ("wqkv", c_void_p), ("bqkv", c_void_p), ("wo", c_void_p),
        ("kcache", c_void_p), ("vcache", c_void_p), ("pos", c_void_p), ("slabs", c_void_p),
        ("kv_batch_stride", c_int64), ("slab_stride", c_int64),
        ("B", C.c_int32), ("d", C.c_int32), ("H", C.c_int32), ("dtype", C.c_int32),
        ("eps", c_float), ("qk_scale", c_float),
    ]


class CrossBlockDesc(C.Structure):
    _fields_ = [
        ("x_in", c_void_p), ("x_out", c_void_p), ("slabs", c_void_p), ("bias_o", c_void_p), ("ln_w", c_void_p), ("ln_b", c_void_p),
        ("wq", c_void_p), ("bq", c_void_p), ("kv", c_void_p), ("out", c_void_p),
        ("slab_stride", c_int64),
        ("n_slabs", C.c_int32), ("B", C.c_int32), ("d", C.c_int32), ("H", C.c_int32), ("Tk", C.c_int32), ("dtype", C.c_int32),
        ("eps", c_float), ("qk_scale", c_float), ("cross_splits", C.c_int32),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("out", c_void_p), ("tk_dev", c_void_p),
        ("q_row_dev", c_void_p), ("lse", c_void_p),
        ("q_bs", c_int64), ("q_rs", c_int64), ("q_hs", c_int64),
        ("k_bs", c_int64), ("k_rs", c_int64), ("k_hs", c_int64),
        ("v_bs", c_int64), ("v_rs", c_int64), ("v_hs", c_int64),
        ("o_bs", c_int64), ("o_rs", c_int64), ("o_hs", c_int64),
        ("B", C.c_int32), ("H", C.c_int32), ("Tq", C.c_int32), ("Tk", C.c_int32),
        ("causal", C.c_int32), ("dtype", C.c_int32),
    ]


class ModelCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
        "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer", "dtype", "f32_split", "dec_w_dtype", "weights_generation",
        "enc_act_fp8", "dec_cross_absorbed", "dec_cross_splits")]


class DecLayout(C.Structure):
    _fields_ = [(n, c_int64) for n in (
        "total_bytes", "tokens", "ld_tok", "pos", "not_done", "sum_logprobs", "logits", "ld_logits",
        "cross_kv", "self_kv", "scratch")]


_P = C.POINTER
# name -> (restype, argtypes).  Must list every function include/wipa.h declares
# (tests/test_abi.py parses the header and compares).
SIGNATURES = {
    "wipa_version": (c_int, []),
    "wipa_last_error": (C.c_char_p, []),
    "wipa_stream_create": (c_int, [C.POINTER(c_void_p)]),
    "wipa_stream_create_cu_limited": (c_int, [c_int, C.POINTER(c_void_p)]),
    "wipa_stream_destroy": (c_int, [c_void_p]),
    "wipa_logmel_tables_bytes": (c_size_t, [c_int]),
    "wipa_logmel_init": (c_int, [c_void_p, c_int, c_void_p]),
    "wipa_logmel_workspace_bytes": (c_size_t, [c_int, c_int]),
    "wipa_logmel": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "wipa_mel_pad_cast": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "wipa_gemm": (c_int, [_P(GemmDesc), c_void_p]),
    "wipa_gemm_dispatch_counts": (c_int, [_P(c_int64), c_int, c_int]),
    "wipa_layernorm": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_int, c_int,
                               c_float, c_void_p]),
    "wipa_layernorm_fp8": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "wipa_rowquant_fp8": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p]),
    "wipa_add_slabs_layernorm": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int64, c_void_p, c_int, c_int64, c_void_p,
                                         c_void_p, c_int, c_int, c_float, c_void_p]),
    "wipa_embed_tokens": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                  c_int, c_void_p]),
    "wipa_attention": (c_int, [_P(AttnDesc), c_void_p]),
    "wipa_flash_attn_enc_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int,
                                       c_int, c_int, c_void_p]),
    "wipa_flash_attn_enc_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int,
                                         c_void_p]),
    "wipa_decode_attn": (c_int, [_P(AttnDesc), c_void_p]),
    "wipa_decode_cross_attn": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "wipa_decode_self_block": (c_int, [_P(SelfBlockDesc), c_void_p]),
    "wipa_decode_cross_block": (c_int, [_P(CrossBlockDesc), c_void_p]),
    "wipa_cross_absorbed_splits": (c_int, [c_int, c_int]),
    "wipa_cross_absorbed_scratch_bytes": (c_size_t, [c_int, c_int, c_int]),
    "wipa_cross_absorbed_init": (c_int, [c_int]),
    "wipa_cross_absorbed_attention": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_size_t,
                                              c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "wipa_cross_absorbed_stream": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "wipa_decode_cross_absorbed_block": (c_int, [_P(CrossBlockDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wipa_decode_cross_absorbed_block_out": (c_int, [_P(CrossBlockDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                                     c_void_p, c_size_t, c_void_p]),
    "wipa_greedy_step": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int,
                                 c_int, c_void_p, c_void_p, c_void_p]),
    "wipa_add_i32": (c_int, [c_void_p, C.c_int32, c_void_p]),
    "wipa_greedy_step_embed": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                       c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "wipa_logits_greedy_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "wipa_logits_greedy_partials_bytes": (c_size_t, [c_int]),
    "wipa_logits_greedy": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                   c_int, c_void_p, c_size_t, c_void_p]),
    "wipa_greedy_step_embed_partials": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                                c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_int, c_int, c_float, c_void_p]),
    "wipa_embed_layernorm": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "wipa_bpe_create": (c_void_p, [c_void_p, c_void_p, c_void_p, c_int]),
    "wipa_bpe_free": (None, [c_void_p]),
    "wipa_bpe_encode_piece": (c_int, [c_void_p, C.c_char_p, c_int, c_void_p, c_int]),
    "wipa_bpe_decode": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int]),
    "wipa_build_token_batch": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, C.c_int32, c_void_p, c_int64]),
    "wipa_profile_begin": (c_int, [c_void_p]),
    "wipa_profile_end": (c_int, [_P(c_float), _P(c_int)]),
    "wipa_encoder_workspace_bytes": (c_size_t, [_P(ModelCfg), c_int]),
    "wipa_encoder_forward": (c_int, [_P(ModelCfg), _P(c_void_p), c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "wipa_decoder_layout": (c_int, [_P(ModelCfg), c_int, _P(DecLayout)]),
    "wipa_decoder_set_audio": (c_int, [_P(ModelCfg), _P(c_void_p), c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "wipa_decoder_begin": (c_int, [_P(ModelCfg), c_void_p, c_size_t, c_int, _P(C.c_int32), c_int, c_void_p]),
    "wipa_decoder_run": (c_int, [_P(ModelCfg), _P(c_void_p), c_void_p, c_size_t, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                 c_int, c_void_p]),
    "wipa_decoder_prefill": (c_int, [_P(ModelCfg), _P(c_void_p), c_void_p, c_size_t, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                     c_void_p]),
    "wipa_decode_cross_attn_multi": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "wipa_decoder_release": (c_int, [c_void_p]),
    "wipa_decoder_logits_workspace_bytes": (c_size_t, [_P(ModelCfg), c_int, c_int]),
    "wipa_decoder_logits": (c_int, [_P(ModelCfg), _P(c_void_p), c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_size_t,
                                    c_int, c_int, c_void_p]),
    "wipa_masked_ce": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                               c_void_p]),
    "wipa_transpose": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "wipa_sum_slabs": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int64, c_int, c_void_p]),
    "wipa_sum_slabs_ex": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int64, c_void_p, c_float, c_void_p]),
    "wipa_colsum": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_void_p, c_int64, c_void_p]),
    "wipa_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                   c_float, c_void_p]),
    "wipa_gelu": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "wipa_gelu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "wipa_masked_ce_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "wipa_embed_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "wipa_attention_bwd": (c_int, [_P(AttnDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                   c_void_p]),
    "wipa_clip_adamw": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                c_void_p, c_void_p, c_void_p, c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                C.c_double, c_void_p]),
}

_lib = None
_lock = threading.Lock()


class WipaError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libwipa.so (once).  Raises if it is not built -- there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise WipaError(
                    f"{LIB_PATH} is missing: the HIP extension is not built "
                    "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path.")
            handle = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(handle, name)  # AttributeError if the symbol is not exported
                fn.restype = res
                fn.argtypes = args
            _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().wipa_last_error()
        raise WipaError(f"{what or 'libwipa'} failed ({rc}): {msg.decode() if msg else ''}")
