"""Thin torch-tensor wrappers over the per-kernel C entry points (include/wipa.h).
Used by the parity tests and by host code that needs a single op; the model
runtime calls the same kernels from C++ (csrc/runtime.hip)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from .runtime import dt_code, on_stream, ptr, sptr


def gemm(A: torch.Tensor, W: torch.Tensor, C_out: torch.Tensor, *, M: int, N: int, K: int, lda: int, ldw: int, ldc: int,
         bias: Optional[torch.Tensor] = None, bias_along_m: bool = False, act: int = 0,
         residual: Optional[torch.Tensor] = None, pos: Optional[torch.Tensor] = None, ldpos: int = 0,
         col_scale_n: int = 0, col_scale: float = 1.0, rg_in: int = 0, rg_valid: int = 0, rg_stride: int = 0,
         cg_in: int = 0, cg_stride: int = 0, c_offset: int = 0, c_offset_dev: Optional[torch.Tensor] = None,
         zero_invalid_rows: bool = False, k_slices: int = 0, slab_stride: int = 0, f32_split: bool = False,
         a_trans: bool = False, w_trans: bool = False) -> torch.Tensor:
    """C = epilogue(A @ W^T); see wipa_gemm in include/wipa.h for the addressing rules.  ``f32_split``: float32 operands
    multiplied as three bf16 MFMA terms (faster, ~5e-6 relative) instead of exact f32 products.  ``a_trans`` / ``w_trans``
    (float32): the operand is stored K-major, [K, M] / [K, N], with lda / ldw its row length."""
    L = _lib.lib()
    d = _lib.GemmDesc()
    d.A, d.W, d.C = ptr(A), ptr(W), ptr(C_out)
    d.bias, d.residual, d.pos, d.c_offset_dev = ptr(bias), ptr(residual), ptr(pos), ptr(c_offset_dev)
    d.lda, d.ldw, d.ldc, d.ldpos = lda, ldw, ldc, ldpos
    d.rg_stride, d.cg_stride, d.c_offset = rg_stride, cg_stride, c_offset
    d.M, d.N, d.K = M, N, K
    d.in_dtype, d.out_dtype = dt_code(A.dtype), dt_code(C_out.dtype)
    assert W.dtype == A.dtype
    d.rg_in, d.rg_valid, d.cg_in = rg_in, rg_valid, cg_in
    d.zero_invalid_rows, d.bias_along_m, d.act = int(zero_invalid_rows), int(bias_along_m), act
    d.col_scale_n, d.col_scale = col_scale_n, col_scale
    d.k_slices, d.slab_stride = k_slices, slab_stride
    d.f32_split = int(f32_split)
    d.a_trans, d.w_trans = int(a_trans), int(w_trans)
    with on_stream() as s:
        _lib.check(L.wipa_gemm(C.byref(d), sptr(s)), "wipa_gemm")
    return C_out


def gemm_fp8(A: torch.Tensor, a_scale: torch.Tensor, W: torch.Tensor, w_scale: torch.Tensor, C_out: torch.Tensor, *,
             bias: Optional[torch.Tensor] = None, act: int = 0, residual: Optional[torch.Tensor] = None, col_scale_n: int = 0,
             col_scale: float = 1.0, bias_along_m: bool = False) -> torch.Tensor:
    """C = epilogue((A * a_scale[:, None]) @ (W * w_scale[:, None])^T) with A [M, K], W [N, K] uint8 OCP e4m3fn codes: the
    fp8 x fp8 tile GEMM on v_mfma_scale_f32_16x16x128_f8f6f4 (wipa_gemm_desc.in_dtype = WIPA_FP8_E4M3)."""
    L = _lib.lib()
    assert A.dtype == torch.uint8 and W.dtype == torch.uint8 and A.shape[1] == W.shape[1]
    d = _lib.GemmDesc()
    d.A, d.W, d.C = ptr(A), ptr(W), ptr(C_out)
    d.bias, d.residual = ptr(bias), ptr(residual)
    d.a_scale, d.w_scale = ptr(a_scale), ptr(w_scale)
    d.lda, d.ldw, d.ldc = A.stride(0), W.stride(0), C_out.stride(0)
    d.M, d.N, d.K = A.shape[0], W.shape[0], A.shape[1]
    d.in_dtype, d.out_dtype = _lib.WIPA_FP8_E4M3, dt_code(C_out.dtype)
    d.act, d.col_scale_n, d.col_scale, d.bias_along_m = act, col_scale_n, col_scale, int(bias_along_m)
    with on_stream() as s:
        _lib.check(L.wipa_gemm(C.byref(d), sptr(s)), "wipa_gemm")
    return C_out


def layernorm_fp8(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5):
    """rows of f32 x -> (e4m3fn codes uint8 [rows, D], power-of-two scale f32 [rows]) of LayerNorm(x): wipa_layernorm_fp8"""
    rows, D = x.shape
    with on_stream() as s:
        y = torch.empty(rows, D, dtype=torch.uint8, device=x.device)
        sc = torch.empty(rows, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().wipa_layernorm_fp8(ptr(x), x.stride(0), ptr(y), D, ptr(sc), ptr(w), ptr(b), rows, D, eps, sptr(s)),
                   "wipa_layernorm_fp8")
    return y, sc


def rowquant_fp8(x: torch.Tensor):
    """bf16 / f32 x [rows, D] -> (e4m3fn codes uint8 [rows, D], power-of-two scale f32 [rows]): wipa_rowquant_fp8"""
    rows, D = x.shape
    with on_stream() as s:
        y = torch.empty(rows, D, dtype=torch.uint8, device=x.device)
        sc = torch.empty(rows, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().wipa_rowquant_fp8(ptr(x), dt_code(x.dtype), x.stride(0), ptr(y), D, ptr(sc), rows, D, sptr(s)),
                   "wipa_rowquant_fp8")
    return y, sc


def gemm_dispatch_counts(reset: bool = False) -> dict:
    """wipa_gemm_dispatch_counts: {kernel family: calls since the last reset} (a test / measurement aid)."""
    n = len(_lib.GEMM_DISPATCH)
    buf = (C.c_int64 * n)()
    _lib.check(_lib.lib().wipa_gemm_dispatch_counts(buf, n, int(reset)), "wipa_gemm_dispatch_counts")
    return dict(zip(_lib.GEMM_DISPATCH, (int(v) for v in buf)))


def linear(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = 0,
           residual: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None, f32_split: bool = False) -> torch.Tensor:
    """x [M,K] @ W[N,K]^T (+bias, gelu, +residual) -> [M,N]."""
    M, K = x.shape
    N = W.shape[0]
    with on_stream():
        out = torch.empty(M, N, dtype=out_dtype or x.dtype, device=x.device)
    return gemm(x, W, out, M=M, N=N, K=K, lda=x.stride(0), ldw=W.stride(0), ldc=N, bias=bias, act=act, residual=residual,
                f32_split=f32_split)


def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, out_dtype: Optional[torch.dtype] = None, eps: float = 1e-5):
    L = _lib.lib()
    rows, D = x.shape
    with on_stream() as s:
        y = torch.empty(rows, D, dtype=out_dtype or x.dtype, device=x.device)
        _lib.check(L.wipa_layernorm(ptr(x), dt_code(x.dtype), x.stride(0), ptr(y), dt_code(y.dtype), D, ptr(w), ptr(b),
                                    rows, D, eps, sptr(s)), "wipa_layernorm")
    return y


def add_slabs_layernorm(x: torch.Tensor, slabs: Optional[torch.Tensor], w: torch.Tensor, b: torch.Tensor,
                        out_dtype: torch.dtype = torch.float32, eps: float = 1e-5) -> torch.Tensor:
    """x [rows, D] f32 (updated in place: += slabs.sum(0) in slab order), slabs [S, rows, D] f32 -> LN(x)."""
    L = _lib.lib()
    rows, D = x.shape
    S = 0 if slabs is None else slabs.shape[0]
    with on_stream() as s:
        y = torch.empty(rows, D, dtype=out_dtype, device=x.device)
        _lib.check(L.wipa_add_slabs_layernorm(ptr(x), x.stride(0), ptr(slabs), S, (slabs.stride(0) if S else 0), ptr(y),
                                              dt_code(out_dtype), D, ptr(w), ptr(b), rows, D, eps, sptr(s)),
                   "wipa_add_slabs_layernorm")
    return y


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, causal: bool = False) -> torch.Tensor:
    """q [B,Tq,H,64], k/v [B,Tk,H,64] (any strides with contiguous last dim) -> [B,Tq,H,64]."""
    L = _lib.lib()
    B, Tq, H, hd = q.shape
    Tk = k.shape[1]
    assert hd == 64 and q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1
    with on_stream() as s:
        out = torch.empty(B, Tq, H, 64, dtype=q.dtype, device=q.device)
        d = _lib.AttnDesc()
        d.q, d.k, d.v, d.out = ptr(q), ptr(k), ptr(v), ptr(out)
        d.q_bs, d.q_rs, d.q_hs = q.stride(0), q.stride(1), q.stride(2)
        d.k_bs, d.k_rs, d.k_hs = k.stride(0), k.stride(1), k.stride(2)
        d.v_bs, d.v_rs, d.v_hs = v.stride(0), v.stride(1), v.stride(2)
        d.o_bs, d.o_rs, d.o_hs = out.stride(0), out.stride(1), out.stride(2)
        d.B, d.H, d.Tq, d.Tk, d.causal, d.dtype = B, H, Tq, Tk, int(causal), dt_code(q.dtype)
        _lib.check(L.wipa_attention(C.byref(d), sptr(s)), "wipa_attention")
    return out


def decode_attn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, Tk: Optional[int] = None,
                tk_dev: Optional[torch.Tensor] = None, q_row_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One query row per (b,h): q [B,Tq_buf,H,64] (row ``*q_row_dev`` or 0 is used), k/v [B,Tk_buf,H,64]
    with any strides (last dim contiguous); keys [0, Tk + *tk_dev) -> [B,H,64]."""
    L = _lib.lib()
    B, _, H, hd = q.shape
    assert hd == 64
    with on_stream() as s:
        out = torch.empty(B, H, 64, dtype=q.dtype, device=q.device)
        d = _lib.AttnDesc()
        d.q, d.k, d.v, d.out = ptr(q), ptr(k), ptr(v), ptr(out)
        d.tk_dev, d.q_row_dev = ptr(tk_dev), ptr(q_row_dev)
        d.q_bs, d.q_rs, d.q_hs = q.stride(0), q.stride(1), q.stride(2)
        d.k_bs, d.k_rs, d.k_hs = k.stride(0), k.stride(1), k.stride(2)
        d.v_bs, d.v_rs, d.v_hs = v.stride(0), v.stride(1), v.stride(2)
        d.o_bs, d.o_rs, d.o_hs = out.stride(0), out.stride(0), out.stride(1)
        d.B, d.H, d.Tq, d.Tk, d.causal, d.dtype = B, H, 1, (k.shape[1] if Tk is None else Tk), 0, dt_code(q.dtype)
        _lib.check(L.wipa_decode_attn(C.byref(d), sptr(s)), "wipa_decode_attn")
    return out


def flash_attn_enc(qk: torch.Tensor, vt: torch.Tensor, B: int, H: int, T: int) -> torch.Tensor:
    """qk [B*T, 2D] bf16 (q|k, pre-scaled), vt [B, D, ldvt] bf16 (zero beyond T) -> [B*T, D] bf16."""
    L = _lib.lib()
    D = H * 64
    with on_stream() as s:
        out = torch.empty(B * T, D, dtype=torch.bfloat16, device=qk.device)
        _lib.check(L.wipa_flash_attn_enc_bf16(ptr(qk), qk.stride(0), ptr(vt), vt.stride(1), ptr(out), D, B, H, T, sptr(s)),
                   "wipa_flash_attn_enc_bf16")
    return out


def flash_attn_enc_f32(qk: torch.Tensor, v: torch.Tensor, B: int, H: int, T: int, f32_split: bool = False) -> torch.Tensor:
    """qk [B*T, 2D] f32 (q|k, pre-scaled), v [B*T, D] f32 -> [B*T, D] f32 on the f32 MFMA (or, with ``f32_split``, on
    bf16 MFMAs over split operands)."""
    L = _lib.lib()
    D = H * 64
    assert qk.dtype == torch.float32 and v.dtype == torch.float32
    with on_stream() as s:
        out = torch.empty(B * T, D, dtype=torch.float32, device=qk.device)
        k = qk[:, D:]
        _lib.check(L.wipa_flash_attn_enc_f32(ptr(qk), qk.stride(0), ptr(k), qk.stride(0), ptr(v), v.stride(0), ptr(out), D,
                                             B, H, T, int(f32_split), sptr(s)), "wipa_flash_attn_enc_f32")
    return out


def decode_cross_attn(q: torch.Tensor, kv: torch.Tensor) -> torch.Tensor:
    """q [B, H*64], kv [B, 2H, Tk, 64] -> [B, H*64]."""
    L = _lib.lib()
    B, twoH, Tk, hd = kv.shape
    assert hd == 64 and kv.is_contiguous() and q.is_contiguous()
    with on_stream() as s:
        out = torch.empty_like(q)
        _lib.check(L.wipa_decode_cross_attn(ptr(q), ptr(kv), ptr(out), B, twoH // 2, Tk, dt_code(q.dtype), sptr(s)),
                   "wipa_decode_cross_attn")
    return out


def embed_tokens(tokens: torch.Tensor, tok_emb: torch.Tensor, pos_emb: torch.Tensor, t_start: int = 0) -> torch.Tensor:
    L = _lib.lib()
    B, T = tokens.shape
    D = tok_emb.shape[1]
    with on_stream() as s:
        x = torch.empty(B * T, D, dtype=torch.float32, device=tok_emb.device)
        _lib.check(L.wipa_embed_tokens(ptr(tokens), tokens.stride(0), B, T, t_start, None, ptr(tok_emb), dt_code(tok_emb.dtype),
                                       None, ptr(pos_emb), ptr(x), D, sptr(s)), "wipa_embed_tokens")
    return x


def masked_ce(logits: torch.Tensor, tokens: torch.Tensor, V: int, eot: int):
    """logits [B*T, ldl] f32 for inputs tokens[:, :-1]; returns (sum masked ce, n valid) as a 2-vector."""
    L = _lib.lib()
    B, Tp1 = tokens.shape
    T = Tp1 - 1
    with on_stream() as s:
        row_buf = torch.empty(2 * B * T, dtype=torch.float32, device=logits.device)
        out = torch.empty(2, dtype=torch.float32, device=logits.device)
        _lib.check(L.wipa_masked_ce(ptr(logits), logits.stride(0), ptr(tokens), tokens.stride(0), B, T, V, eot, ptr(row_buf),
                                    ptr(out), sptr(s)), "wipa_masked_ce")
    return out, row_buf
