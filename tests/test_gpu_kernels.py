"""GPU parity tests, kernel level: every HIP kernel through the C ABI (ctypes) against a plain
torch fp32 statement of the same op.  Run with ``pytest -m gpu`` on an MI355X."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from whisper_ipa_amd import ops as o

    return o


def _rel(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 96), (64, 768, 768), (1, 51865, 128), (1000, 402, 416)])
def test_gemm_f32_plain(ops, M, N, K, f32_mode):
    tol = 2e-6 if f32_mode == "exact" else 1e-5  # f32 MFMA vs three-term bf16 split (tile kernels only)
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g)
    ldc = (N + 7) // 8 * 8
    out = torch.full((M, ldc), 7.0, device="cuda")
    ops.gemm(A.cuda(), W.cuda(), out, M=M, N=N, K=K, lda=K, ldw=K, ldc=ldc, f32_split=(f32_mode == "split"))
    torch.cuda.synchronize()
    ref = A.double() @ W.double().T
    assert _rel(out[:, :N], ref) < tol
    if ldc > N:
        assert (out[:, N:] == 7.0).all()  # padding columns untouched


@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_bf16_epilogue(ops, out_dtype):
    g = torch.Generator().manual_seed(3)
    M, N, K = 333, 256, 192
    A = torch.randn(M, K, generator=g).bfloat16()
    W = (torch.randn(N, K, generator=g) * 0.1).bfloat16()
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g).to(out_dtype)
    out = res.clone().cuda()
    ops.gemm(A.cuda(), W.cuda(), out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, bias=bias.cuda(), act=1, residual=out,
             col_scale_n=100, col_scale=0.5)
    torch.cuda.synchronize()
    y = A.float() @ W.float().T + bias
    y[:, :100] *= 0.5
    ref = torch.nn.functional.gelu(y) + res.float()
    tol = 1e-2 if out_dtype == torch.bfloat16 else 2e-5
    assert _rel(out, ref) < tol


def test_gemm_overlapping_rows_and_remap(ops):
    """conv1d(k=3, pad=1) as a GEMM over a halo-padded, row-overlapping A, output shifted by one
    row with zeroed dead rows -- the layout trick of the encoder front end."""
    g = torch.Generator().manual_seed(5)
    B, L, Cin, Cout = 2, 50, 32, 64
    x = torch.randn(B, L, Cin, generator=g)
    w = torch.randn(Cout, 3, Cin, generator=g) * 0.2
    bias = torch.randn(Cout, generator=g)
    P = L + 2
    xp = torch.zeros(B * P + 4, Cin)
    for b in range(B):
        xp[b * P + 1 : b * P + 1 + L] = x[b]
    out = torch.full((B * P + 4, Cout), 9.0)
    out[0] = 0
    out = out.cuda()
    ops.gemm(xp.cuda(), w.reshape(Cout, 3 * Cin).contiguous().cuda(), out, M=B * P, N=Cout, K=3 * Cin, lda=Cin, ldw=3 * Cin,
             ldc=Cout, bias=bias.cuda(), rg_in=P, rg_valid=L, rg_stride=P * Cout, zero_invalid_rows=True, c_offset=Cout)
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv1d(x.transpose(1, 2), w.permute(0, 2, 1), bias, padding=1).transpose(1, 2)
    got = out.cpu()
    for b in range(B):
        assert _rel(got[b * P + 1 : b * P + 1 + L], ref[b]) < 1e-5
        assert (got[b * P] == 0).all() and (got[b * P + L + 1] == 0).all()


def test_gemm_transposed_output_groups(ops):
    """bias along M + column groups: the V^T-per-clip layout of the encoder."""
    g = torch.Generator().manual_seed(6)
    d, B, T, Tp = 64, 3, 20, 32
    Wv = torch.randn(d, d, generator=g)
    x = torch.randn(B * T, d, generator=g)
    bv = torch.randn(d, generator=g)
    out = torch.zeros(B, d, Tp).cuda()
    ops.gemm(Wv.cuda(), x.cuda(), out, M=d, N=B * T, K=d, lda=d, ldw=d, ldc=Tp, bias=bv.cuda(), bias_along_m=True, cg_in=T,
             cg_stride=d * Tp)
    torch.cuda.synchronize()
    ref = (x @ Wv.T + bv).view(B, T, d).transpose(1, 2)
    assert _rel(out[:, :, :T], ref) < 1e-5
    assert (out[:, :, T:] == 0).all()


@pytest.mark.parametrize("din,dout", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("D", [128, 384, 768, 1280])
def test_layernorm(ops, din, dout, D):
    g = torch.Generator().manual_seed(D)
    x = (torch.randn(37, D, generator=g) * 3 + 1).to(din)
    w = torch.randn(D, generator=g)
    b = torch.randn(D, generator=g)
    y = ops.layernorm(x.cuda(), w.cuda(), b.cuda(), out_dtype=dout)
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x.float(), (D,), w, b, 1e-5)
    assert _rel(y, ref) < (1e-2 if dout == torch.bfloat16 else 1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Tq,Tk,causal", [(37, 150, False), (21, 21, True), (1, 300, False), (5, 69, True), (130, 1500, False)])
def test_attention_generic(ops, dtype, Tq, Tk, causal):
    g = torch.Generator().manual_seed(Tq * 1000 + Tk)
    B, H = 2, 3
    q = (torch.randn(B, Tq, H, 64, generator=g) * 0.5).to(dtype)
    k = (torch.randn(B, Tk, H, 64, generator=g) * 0.5).to(dtype)
    v = torch.randn(B, Tk, H, 64, generator=g).to(dtype)
    out = ops.attention(q.cuda(), k.cuda(), v.cuda(), causal=causal)
    torch.cuda.synchronize()
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float())
    if causal:
        mask = torch.triu(torch.full((Tk, Tk), float("-inf")), 1)[Tk - Tq :]
        s = s + mask
    ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), v.float())
    assert _rel(out, ref) < (1e-2 if dtype == torch.bfloat16 else 1e-5)


@pytest.mark.parametrize("T", [1500, 128, 200])
def test_flash_attention_encoder_bf16(ops, T):
    g = torch.Generator().manual_seed(T)
    B, H = 2, 2
    D = H * 64
    q = (torch.randn(B, T, H, 64, generator=g) * 0.6).bfloat16()
    k = (torch.randn(B, T, H, 64, generator=g) * 0.6).bfloat16()
    v = torch.randn(B, T, H, 64, generator=g).bfloat16()
    qk = torch.cat([q.reshape(B * T, D), k.reshape(B * T, D)], dim=1).contiguous()
    Tp = (T + 63) // 64 * 64
    vt = torch.zeros(B, D, Tp, dtype=torch.bfloat16)
    vt[:, :, :T] = v.reshape(B, T, D).transpose(1, 2)
    out = ops.flash_attn_enc(qk.cuda(), vt.cuda(), B, H, T)
    torch.cuda.synchronize()
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float())
    ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), v.float()).reshape(B * T, D)
    assert _rel(out, ref) < 1.5e-2


def test_flash_attention_spiked_scores(ops):
    """force large running-max jumps between key tiles (online-softmax rescale path)."""
    g = torch.Generator().manual_seed(11)
    B, H, T = 1, 1, 256
    q = (torch.randn(B, T, H, 64, generator=g) * 0.3).bfloat16()
    k = (torch.randn(B, T, H, 64, generator=g) * 0.3).bfloat16()
    v = torch.randn(B, T, H, 64, generator=g).bfloat16()
    for t in (70, 140, 250):  # later tiles hold much larger scores for some queries
        k[0, t, 0] = (q[0, t % 37, 0].float() * (t / 10.0)).bfloat16()
    qk = torch.cat([q.reshape(T, 64), k.reshape(T, 64)], dim=1).contiguous()
    vt = v.reshape(1, T, 64).transpose(1, 2).contiguous()
    out = ops.flash_attn_enc(qk.cuda(), vt.cuda(), B, H, T)
    torch.cuda.synchronize()
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float())
    ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), v.float()).reshape(T, 64)
    assert _rel(out, ref) < 1.5e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n_q", [2, 3, 4])
def test_decode_cross_attention_multi_query(dtype, n_q):
    """prompt prefill: n_q query rows per clip against one pass over the cached cross K/V == n_q single-query launches."""
    import ctypes as C

    from whisper_ipa_amd import _lib, ops as O
    from whisper_ipa_amd.runtime import dt_code, on_stream, ptr, sptr

    g = torch.Generator().manual_seed(n_q)
    B, H, Tk = 3, 2, 1500
    q = (torch.randn(B * n_q, H * 64, generator=g) * 0.5).to(dtype).cuda()
    kv = torch.randn(B, 2 * H, Tk, 64, generator=g).to(dtype).cuda()
    out = torch.empty_like(q)
    with on_stream() as s:
        _lib.check(_lib.lib().wipa_decode_cross_attn_multi(ptr(q), ptr(kv), ptr(out), B, H, Tk, n_q, dt_code(dtype), sptr(s)))
    torch.cuda.synchronize()
    for t in range(n_q):
        one = O.decode_cross_attn(q.view(B, n_q, -1)[:, t].contiguous(), kv)
        torch.cuda.synchronize()
        assert torch.equal(out.view(B, n_q, -1)[:, t], one), t


@pytest.mark.parametrize("T", [1500, 200, 64, 37])
def test_flash_attention_encoder_f32(ops, T, f32_mode):
    """float32 flash attention (the reference's own dtype) against softmax(QK^T)V in float64; ragged last key tile,
    partial query block, and spiked scores that move the running maximum between tiles.  Exact mode (default): f32 MFMA kernel;
    opt-in: three-term bf16 split of every product (score errors of ~1e-5 relative pass through the exponential)."""
    tol = 2e-5 if f32_mode == "exact" else 1e-4
    g = torch.Generator().manual_seed(T)
    B, H = 2, 3
    D = H * 64
    q = torch.randn(B, T, H, 64, generator=g) * 0.6
    k = torch.randn(B, T, H, 64, generator=g) * 0.6
    v = torch.randn(B, T, H, 64, generator=g)
    for t in range(5, T, 61):
        k[0, t, 1] = q[0, t % 29, 1] * (1.0 + t / 40.0)
    qk = torch.cat([q.reshape(B * T, D), k.reshape(B * T, D)], dim=1).contiguous()
    out = ops.flash_attn_enc_f32(qk.cuda(), v.reshape(B * T, D).contiguous().cuda(), B, H, T, f32_split=(f32_mode == "split"))
    torch.cuda.synchronize()
    s = torch.einsum("bqhd,bkhd->bhqk", q.double(), k.double())
    ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), v.double()).reshape(B * T, D).float()
    assert _rel(out, ref) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Tk", [1500, 97, 8])
def test_decode_cross_attention(ops, dtype, Tk):
    g = torch.Generator().manual_seed(Tk)
    B, H = 3, 2
    q = (torch.randn(B, H * 64, generator=g) * 0.5).to(dtype)
    kv = torch.randn(B, 2 * H, Tk, 64, generator=g).to(dtype)
    out = ops.decode_cross_attn(q.cuda(), kv.cuda())
    torch.cuda.synchronize()
    qf = q.float().view(B, H, 64)
    K, V = kv[:, :H].float(), kv[:, H:].float()
    s = torch.einsum("bhd,bhtd->bht", qf, K)
    ref = torch.einsum("bht,bhtd->bhd", torch.softmax(s, -1), V).reshape(B, H * 64)
    assert _rel(out, ref) < (1e-2 if dtype == torch.bfloat16 else 1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_decode_self_attention_cache_layout(ops, dtype):
    """self-attention cache [B, n_ctx, H*64] with the position in device memory."""
    g = torch.Generator().manual_seed(9)
    B, H, nctx, pos = 3, 2, 448, 70
    qbuf = (torch.randn(B, nctx, H, 64, generator=g) * 0.5).to(dtype)
    kbuf = (torch.randn(B, nctx, H, 64, generator=g) * 0.5).to(dtype)
    vbuf = torch.randn(B, nctx, H, 64, generator=g).to(dtype)
    pos_dev = torch.tensor([pos], dtype=torch.int32).cuda()
    out = ops.decode_attn(qbuf.cuda(), kbuf.cuda(), vbuf.cuda(), Tk=1, tk_dev=pos_dev, q_row_dev=pos_dev)
    torch.cuda.synchronize()
    qf = qbuf[:, pos].float()
    K, V = kbuf[:, : pos + 1].float(), vbuf[:, : pos + 1].float()
    s = torch.einsum("bhd,bthd->bht", qf, K)
    ref = torch.einsum("bht,bthd->bhd", torch.softmax(s, -1), V)
    assert _rel(out, ref) < (1e-2 if dtype == torch.bfloat16 else 1e-5)


@pytest.mark.parametrize("dtype,out_dtype", [(torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32), (torch.float32, torch.float32)])
@pytest.mark.parametrize("M,N,K", [(64, 768, 768), (64, 768, 3072), (64, 2304, 768), (1, 768, 768), (17, 100, 128), (33, 51865, 768), (28, 384, 1536)])
def test_gemm_skinny(ops, dtype, out_dtype, M, N, K):
    """decode-step GEMMs (M <= 64) take the weight-streaming kernel: all epilogue features."""
    g = torch.Generator().manual_seed(M * 7 + N + K)
    A = (torch.randn(M, K, generator=g)).to(dtype)
    W = (torch.randn(N, K, generator=g) * 0.05).to(dtype)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g).to(out_dtype)
    ldc = (N + 7) // 8 * 8
    out = torch.zeros(M, ldc, dtype=out_dtype)
    out[:, :N] = res
    out = out.cuda()
    ops.gemm(A.cuda(), W.cuda(), out, M=M, N=N, K=K, lda=K, ldw=K, ldc=ldc, bias=bias.cuda(), act=1, residual=out,
             col_scale_n=N // 2, col_scale=0.25)
    torch.cuda.synchronize()
    y = A.double() @ W.double().T + bias.double()
    y[:, : N // 2] *= 0.25
    ref = torch.nn.functional.gelu(y) + res.double()
    tol = 1e-2 if out_dtype == torch.bfloat16 else (3e-5 if dtype == torch.float32 else 1e-5)
    assert _rel(out[:, :N], ref) < tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K,S", [(64, 768, 768, 2), (64, 768, 3072, 4), (7, 128, 256, 3), (64, 768, 768, 1)])
def test_gemm_split_k_slabs_and_fused_layernorm(ops, dtype, M, N, K, S):
    """deterministic split-K: slice z writes a partial slab (bias in slab 0); the fused kernel adds the
    slabs to the residual stream in order and applies LayerNorm."""
    g = torch.Generator().manual_seed(M + N + K + S)
    A = torch.randn(M, K, generator=g).to(dtype)
    W = (torch.randn(N, K, generator=g) * 0.05).to(dtype)
    bias = torch.randn(N, generator=g)
    x0 = torch.randn(M, N, generator=g)
    lw, lb = torch.randn(N, generator=g), torch.randn(N, generator=g)
    slabs = torch.full((S, M, N), float("nan")).cuda()
    ops.gemm(A.cuda(), W.cuda(), slabs, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, bias=bias.cuda(), k_slices=S, slab_stride=M * N)
    x = x0.clone().cuda()
    y = ops.add_slabs_layernorm(x, slabs, lw.cuda(), lb.cuda(), out_dtype=torch.float32)
    y2 = ops.add_slabs_layernorm(x.clone(), None, lw.cuda(), lb.cuda(), out_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    ref_x = x0.double() + A.double() @ W.double().T + bias.double()
    assert _rel(slabs.sum(0), ref_x - x0.double()) < 2e-5
    assert _rel(x, ref_x) < 2e-5
    ref_y = torch.nn.functional.layer_norm(ref_x.float(), (N,), lw, lb, 1e-5)
    assert _rel(y, ref_y) < 5e-5
    assert _rel(y2, ref_y) < 1e-2
    # run-to-run determinism (no atomics)
    slabs2 = torch.zeros_like(slabs)
    ops.gemm(A.cuda(), W.cuda(), slabs2, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, bias=bias.cuda(), k_slices=S, slab_stride=M * N)
    torch.cuda.synchronize()
    assert torch.equal(slabs, slabs2)


def test_embed_tokens(ops):
    g = torch.Generator().manual_seed(1)
    V, D, B, T = 1000, 128, 3, 7
    emb = torch.randn(V, D, generator=g)
    pos = torch.randn(448, D, generator=g)
    tok = torch.randint(0, V, (B, T), generator=g, dtype=torch.int32)
    x = ops.embed_tokens(tok.cuda(), emb.cuda(), pos.cuda())
    torch.cuda.synchronize()
    ref = emb[tok.long()] + pos[:T]
    assert torch.equal(x.cpu().view(B, T, D), ref)


def test_masked_ce(ops):
    g = torch.Generator().manual_seed(2)
    B, T, V, eot = 3, 9, 51865, 50257
    logits = torch.randn(B * T, V + 7, generator=g) * 2
    tokens = torch.randint(0, 50000, (B, T + 1), generator=g, dtype=torch.int32)
    tokens[0, 6:] = eot
    tokens[1, 3:] = eot
    out, rows = ops.masked_ce(logits.cuda(), tokens.cuda(), V, eot)
    torch.cuda.synchronize()
    tgt = tokens[:, 1:].long()
    is_eot = tgt == eot
    mask = (~is_eot) | (torch.cumsum(is_eot.long(), 1) == 1)
    ce = torch.nn.functional.cross_entropy(logits[:, :V], tgt.reshape(-1), reduction="none")
    ref_sum = float((ce * mask.reshape(-1)).sum())
    assert abs(float(out[0]) - ref_sum) / ref_sum < 1e-5
    assert int(out[1]) == int(mask.sum())


# ------------------------------------------------------------------ fused decode-step blocks (csrc/decode_fused.hip)
def _ln_ref(x, w, b, eps=1e-5):
    x = x.double()
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w.double() + b.double()


def _rt(x, dtype):
    """round through the storage dtype, back to float64"""
    return x.to(torch.float32).to(dtype).double()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,pos", [(5, 2, 0), (16, 2, 7), (37, 3, 70), (64, 12, 200), (3, 20, 447)])
def test_decode_self_block(dtype, B, H, pos):
    """LayerNorm -> q|k|v of one head -> cache append -> self-attention -> out-projection slab, against a float64 statement
    with the same rounding points (LN output, q, k, v and the attention output are stored as T)."""
    import ctypes as C

    from whisper_ipa_amd import _lib
    from whisper_ipa_amd.runtime import dt_code, on_stream, ptr, sptr

    L = _lib.lib()
    d, nctx = H * 64, 448
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + pos)
    x = torch.randn(B, d, generator=g) * 1.5
    ln_w, ln_b = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    wqkv = (torch.randn(3 * d, d, generator=g) * 0.06).to(dtype)
    bqkv = torch.randn(3 * d, generator=g) * 0.1
    bqkv[d:2 * d] = 0
    wo = (torch.randn(d, d, generator=g) * 0.06).to(dtype)
    kc = (torch.randn(B, nctx, d, generator=g) * 0.5).to(dtype)
    vc = torch.randn(B, nctx, d, generator=g).to(dtype)
    kc[:, pos:] = float("nan")  # slots from `pos` on are unwritten memory in real use: nothing of them may reach the result
    vc[:, pos:] = float("nan")
    scale = 64 ** -0.25
    with on_stream() as s:
        dev = [t.cuda() for t in (x, ln_w, ln_b, wqkv, bqkv, wo, kc, vc)]
        xd, lwd, lbd, wqkvd, bqkvd, wod, kcd, vcd = dev
        posd = torch.tensor([pos], dtype=torch.int32, device="cuda")
        slabs = torch.full((H, B, d), 7.0, device="cuda")
        a = _lib.SelfBlockDesc()
        a.x, a.ln_w, a.ln_b, a.wqkv, a.bqkv, a.wo = ptr(xd), ptr(lwd), ptr(lbd), ptr(wqkvd), ptr(bqkvd), ptr(wod)
        a.kcache, a.vcache, a.pos, a.slabs = ptr(kcd), ptr(vcd), ptr(posd), ptr(slabs)
        a.kv_batch_stride, a.slab_stride = nctx * d, B * d
        a.B, a.d, a.H, a.dtype, a.eps, a.qk_scale = B, d, H, dt_code(dtype), 1e-5, scale
        _lib.check(L.wipa_decode_self_block(C.byref(a), sptr(s)), "wipa_decode_self_block")
    torch.cuda.synchronize()
    y = _rt(_ln_ref(x, ln_w, ln_b), dtype)
    qkv = y @ wqkv.double().T + bqkv.double()
    q, k, v = _rt(qkv[:, :d] * scale, dtype), _rt(qkv[:, d:2 * d] * scale, dtype), _rt(qkv[:, 2 * d:], dtype)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    # the new K/V row sits at `pos`, everything else in the caches is untouched
    assert _rel(kcd[:, pos], k) < tol and _rel(vcd[:, pos], v) < tol
    keep = torch.ones(nctx, dtype=torch.bool)
    keep[pos] = False
    assert torch.equal(kcd.cpu()[:, :pos], kc[:, :pos]) and torch.equal(vcd.cpu()[:, :pos], vc[:, :pos])
    assert torch.isnan(kcd[:, pos + 1:].float()).all() and torch.isnan(vcd[:, pos + 1:].float()).all()  # untouched
    K = torch.cat([kc[:, :pos].double(), k[:, None]], 1).view(B, pos + 1, H, 64)
    V = torch.cat([vc[:, :pos].double(), v[:, None]], 1).view(B, pos + 1, H, 64)
    sc = torch.einsum("bhd,bthd->bht", q.view(B, H, 64), K)
    o = _rt(torch.einsum("bht,bthd->bhd", torch.softmax(sc, -1), V), dtype)  # [B, H, 64]
    ref = torch.einsum("bhj,hnj->hbn", o, wo.double().view(d, H, 64).permute(1, 0, 2))  # slab h = o_h @ Wo[:, h]^T
    assert _rel(slabs, ref) < (2e-5 if dtype == torch.float32 else 3e-2), _rel(slabs, ref)
    # the sum over heads is the whole out projection
    full = o.reshape(B, d) @ wo.double().T
    assert _rel(slabs.double().sum(0), full) < (2e-5 if dtype == torch.float32 else 3e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,Tk", [(3, 2, 1500), (64, 12, 1500), (5, 20, 97), (17, 6, 8)])
def test_decode_cross_block(dtype, B, H, Tk):
    """residual + bias + head slabs -> LayerNorm -> cross query -> streaming cross-attention, against float64."""
    import ctypes as C

    from whisper_ipa_amd import _lib
    from whisper_ipa_amd.runtime import dt_code, on_stream, ptr, sptr

    L = _lib.lib()
    d = H * 64
    g = torch.Generator().manual_seed(B * 100 + H + Tk)
    x = torch.randn(B, d, generator=g)
    slabs = torch.randn(H, B, d, generator=g) * 0.3
    bo = torch.randn(d, generator=g) * 0.1
    ln_w, ln_b = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    wq = (torch.randn(d, d, generator=g) * 0.06).to(dtype)
    bq = torch.randn(d, generator=g) * 0.1
    kv = torch.randn(B, 2 * H, Tk, 64, generator=g).to(dtype)
    kv[:, :H] *= 0.5
    scale = 64 ** -0.25
    with on_stream() as s:
        xd, sd, bod, lwd, lbd, wqd, bqd, kvd = [t.cuda() for t in (x, slabs, bo, ln_w, ln_b, wq, bq, kv)]
        x_out = torch.full((B, d), 7.0, device="cuda")
        out = torch.zeros(B, d, device="cuda", dtype=dtype)
        c = _lib.CrossBlockDesc()
        c.x_in, c.x_out, c.slabs, c.bias_o, c.ln_w, c.ln_b = ptr(xd), ptr(x_out), ptr(sd), ptr(bod), ptr(lwd), ptr(lbd)
        c.wq, c.bq, c.kv, c.out = ptr(wqd), ptr(bqd), ptr(kvd), ptr(out)
        c.slab_stride = B * d
        c.n_slabs, c.B, c.d, c.H, c.Tk, c.dtype, c.eps, c.qk_scale = H, B, d, H, Tk, dt_code(dtype), 1e-5, scale
        _lib.check(L.wipa_decode_cross_block(C.byref(c), sptr(s)), "wipa_decode_cross_block")
        # aliasing the residual buffers is refused (H workgroups read the row that one of them writes)
        c.x_out = c.x_in
        assert L.wipa_decode_cross_block(C.byref(c), sptr(s)) != 0
    torch.cuda.synchronize()
    r = x.double() + bo.double() + slabs.double().sum(0)
    assert _rel(x_out, r) < 2e-6
    y = _rt(_ln_ref(r, ln_w, ln_b), dtype)
    q = _rt((y @ wq.double().T + bq.double()) * scale, dtype).view(B, H, 64)
    Kc, Vc = kv[:, :H].double(), kv[:, H:].double()
    sc = torch.einsum("bhd,bhtd->bht", q, Kc)
    ref = torch.einsum("bht,bhtd->bhd", torch.softmax(sc, -1), Vc).reshape(B, d)
    assert _rel(out, ref) < (2e-5 if dtype == torch.float32 else 2e-2), _rel(out, ref)


@pytest.mark.parametrize("B,H,Tk,n_slabs", [(64, 12, 1500, 2), (5, 6, 1500, 1), (3, 12, 200, 4), (2, 8, 40, 2), (7, 12, 191, 3)])
def test_decode_cross_block_lds_staged_stream(monkeypatch, B, H, Tk, n_slabs):
    """Round 3: the cross block whose first K / V rows travel to LDS by LDS-DMA under the prologue
    (decode_cross_block_pre_kernel, WIPA_CROSS_PRE = 4 / 6; default for grids that are resident at once).  Against float64,
    against the plain kernel (WIPA_CROSS_PRE=0: bit-identical with 4 row groups per step -- the same summation order -- and
    equal up to the softmax grouping with 6), ragged key counts (Tk not a multiple of a step, shorter than one step of all
    four waves), 1..4 slabs."""
    import ctypes as C

    from whisper_ipa_amd import _lib
    from whisper_ipa_amd.runtime import dt_code, on_stream, ptr, sptr

    L = _lib.lib()
    dtype = torch.bfloat16
    d = H * 64
    g = torch.Generator().manual_seed(B * 100 + H + Tk)
    x = torch.randn(B, d, generator=g)
    slabs = torch.randn(n_slabs, B, d, generator=g) * 0.3
    ln_w, ln_b = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    wq = (torch.randn(d, d, generator=g) * 0.06).to(dtype)
    bq = torch.randn(d, generator=g) * 0.1
    kv = torch.randn(B, 2 * H, Tk, 64, generator=g).to(dtype)
    kv[:, :H] *= 0.5
    scale = 64 ** -0.25
    outs = {}
    with on_stream() as s:
        xd, sd, lwd, lbd, wqd, bqd, kvd = [t.cuda() for t in (x, slabs, ln_w, ln_b, wq, bq, kv)]
        for pre in ("0", "4", "6"):
            monkeypatch.setenv("WIPA_CROSS_PRE", pre)
            x_out = torch.full((B, d), 7.0, device="cuda")
            out = torch.zeros(B, d, device="cuda", dtype=dtype)
            c = _lib.CrossBlockDesc()
            c.x_in, c.x_out, c.slabs, c.bias_o, c.ln_w, c.ln_b = ptr(xd), ptr(x_out), ptr(sd), None, ptr(lwd), ptr(lbd)
            c.wq, c.bq, c.kv, c.out = ptr(wqd), ptr(bqd), ptr(kvd), ptr(out)
            c.slab_stride = B * d
            c.n_slabs, c.B, c.d, c.H, c.Tk, c.dtype, c.eps, c.qk_scale = n_slabs, B, d, H, Tk, dt_code(dtype), 1e-5, scale
            _lib.check(L.wipa_decode_cross_block(C.byref(c), sptr(s)), "wipa_decode_cross_block")
            outs[pre] = (x_out, out)
    torch.cuda.synchronize()
    r = x.double() + slabs.double().sum(0)
    y = _rt(_ln_ref(r, ln_w, ln_b), dtype)
    q = _rt((y @ wq.double().T + bq.double()) * scale, dtype).view(B, H, 64)
    Kc, Vc = kv[:, :H].double(), kv[:, H:].double()
    sc = torch.einsum("bhd,bhtd->bht", q, Kc)
    ref = torch.einsum("bht,bhtd->bhd", torch.softmax(sc, -1), Vc).reshape(B, d)
    for pre, (x_out, out) in outs.items():
        assert _rel(x_out, r) < 2e-6, pre
        assert _rel(out, ref) < 2e-2, (pre, _rel(out, ref))
    assert torch.equal(outs["4"][0], outs["0"][0]) and torch.equal(outs["6"][0], outs["0"][0])  # the residual row: same adds
    assert torch.equal(outs["4"][1], outs["0"][1])       # same query, same row groups, same order: bit-identical
    assert _rel(outs["6"][1], outs["0"][1]) < 1e-2       # other softmax grouping: equal up to bf16 rounding of the output


@pytest.mark.parametrize("in_dtype,out_dtype", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16),
                                                (torch.bfloat16, torch.float32)])
@pytest.mark.parametrize("M,N,K,act", [(64, 3072, 768, 1), (5, 200, 128, 0), (100, 1536, 384, 1), (256, 4096, 1024, 1),
                                       (33, 640, 1280, 0)])
def test_gemm_layernorm_prologue(ops, in_dtype, out_dtype, M, N, K, act):
    """wipa_gemm with ln_x: A = LayerNorm(rows of the f32 residual stream) computed inside the weight-streaming kernel."""
    import ctypes as C

    from whisper_ipa_amd import _lib
    from whisper_ipa_amd.runtime import dt_code, on_stream, ptr, sptr

    L = _lib.lib()
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K + 8, generator=g)[:, :K] * 2 + 0.3  # row stride K + 8
    ln_w, ln_b = 1 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    W = (torch.randn(N, K, generator=g) * 0.05).to(in_dtype)
    bias = torch.randn(N, generator=g) * 0.1
    with on_stream() as s:
        xd = torch.empty(M, K + 8, device="cuda")
        xd[:, :K] = x.cuda()
        lwd, lbd, Wd, bd = ln_w.cuda(), ln_b.cuda(), W.cuda(), bias.cuda()
        out = torch.full((M, N), 7.0, device="cuda", dtype=out_dtype)
        dsc = _lib.GemmDesc()
        dsc.W, dsc.C, dsc.bias = ptr(Wd), ptr(out), ptr(bd)
        dsc.ln_x, dsc.ln_w, dsc.ln_b, dsc.ln_ldx, dsc.ln_eps = ptr(xd), ptr(lwd), ptr(lbd), K + 8, 1e-5
        dsc.lda, dsc.ldw, dsc.ldc = K, K, N
        dsc.M, dsc.N, dsc.K, dsc.in_dtype, dsc.out_dtype, dsc.act = M, N, K, dt_code(in_dtype), dt_code(out_dtype), act
        _lib.check(L.wipa_gemm(C.byref(dsc), sptr(s)), "wipa_gemm(ln prologue)")
    torch.cuda.synchronize()
    y = _rt(_ln_ref(x, ln_w, ln_b), in_dtype)
    ref = y @ W.double().T + bias.double()
    if act:
        ref = torch.nn.functional.gelu(ref)
    tol = 1e-5 if in_dtype == torch.float32 else (2e-2 if out_dtype == torch.bfloat16 else 5e-3)
    assert _rel(out, ref) < tol, _rel(out, ref)


# ------------------------------------------------------------------ fp8 (e4m3) weights
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K,act,slices", [(64, 768, 768, 0, 1), (64, 3072, 768, 1, 1), (64, 768, 3072, 0, 4), (1, 51866, 1280, 0, 1),
                                              (130, 200, 128, 0, 1), (37, 1280, 5120, 0, 3), (256, 768, 768, 0, 2)])
def test_gemm_fp8_weights(ops, out_dtype, M, N, K, act, slices):
    """wipa_gemm with w_dtype = WIPA_FP8_E4M3: bf16 activations x e4m3 codes * per-row scale, against float64 on the
    dequantised weights (fp8 -> bf16 is exact and the power-of-two scale commutes with the f32 accumulation, so the error is
    that of a bf16 GEMM)."""
    import ctypes as C

    from whisper_ipa_amd import _lib
    from whisper_ipa_amd.runtime import dt_code, on_stream, ptr, sptr
    from whisper_ipa_amd.whisper import dequantize_fp8_e4m3, quantize_fp8_e4m3

    L = _lib.lib()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).bfloat16()
    W = torch.randn(N, K, generator=g) * torch.logspace(-2, 0, N)[:, None]
    codes, scale = quantize_fp8_e4m3(W)
    Wdq = dequantize_fp8_e4m3(codes, scale)
    bias = torch.randn(N, generator=g) * 0.1
    if slices > 1 and out_dtype != torch.float32:
        pytest.skip("split-K slabs are f32")
    with on_stream() as s:
        Ad, cd, sd, bd = A.cuda(), codes.cuda(), scale.cuda(), bias.cuda()
        ldc = (N + 7) // 8 * 8
        out = torch.full((max(slices, 1), M, ldc), 7.0, device="cuda", dtype=out_dtype)
        dsc = _lib.GemmDesc()
        dsc.A, dsc.W, dsc.C, dsc.bias = ptr(Ad), ptr(cd), ptr(out), ptr(bd)
        dsc.w_scale, dsc.w_dtype = ptr(sd), _lib.WIPA_FP8_E4M3
        dsc.lda, dsc.ldw, dsc.ldc = K, K, ldc
        dsc.M, dsc.N, dsc.K, dsc.in_dtype, dsc.out_dtype, dsc.act = M, N, K, dt_code(torch.bfloat16), dt_code(out_dtype), act
        dsc.k_slices, dsc.slab_stride, dsc.stream_weights = slices, M * ldc, 1
        _lib.check(L.wipa_gemm(C.byref(dsc), sptr(s)), "wipa_gemm(fp8)")
        # rows beyond the weight-streaming kernel are refused (the tile kernels run on dequantised weights)
        dsc.M = 2048
        assert L.wipa_gemm(C.byref(dsc), sptr(s)) != 0
    torch.cuda.synchronize()
    ref = A.double() @ Wdq.double().T + bias.double()
    if act:
        ref = torch.nn.functional.gelu(ref)
    got = out.double().sum(0)[:, :N] if slices > 1 else out[0, :, :N]
    assert _rel(got, ref) < (1e-2 if out_dtype == torch.bfloat16 else 1e-5), _rel(got, ref)
    if ldc > N:
        assert (out[..., N:] == 7.0).all()


def test_embed_fp8_decodes_every_code_like_the_format_definition():
    """The hardware conversion the fp8 kernels rely on is OCP e4m3fn on gfx950 (not MI300's fnuz): all 254 finite codes
    through the fp8 embedding kernel against the bit-level definition."""
    from whisper_ipa_amd import _lib
    from whisper_ipa_amd.runtime import on_stream, ptr, sptr

    L = _lib.lib()
    vals = []
    for c in range(256):
        sgn, e, m = c >> 7, (c >> 3) & 15, c & 7
        v = float("nan") if (e == 15 and m == 7) else (2.0 ** -6 * m / 8 if e == 0 else 2.0 ** (e - 7) * (1 + m / 8))
        vals.append(-v if sgn else v)
    table = torch.tensor(vals, dtype=torch.float32)
    codes = torch.arange(256, dtype=torch.uint8).view(4, 64)  # 4 "token" rows of 64 dims
    scale = torch.tensor([1.0, 0.5, 4.0, 2.0 ** -10])
    with on_stream() as s:
        tok = torch.tensor([[3, 0, 2, 1]], dtype=torch.int32, device="cuda")
        pos = torch.zeros(8, 64, device="cuda")
        x = torch.empty(4, 64, device="cuda")
        cd, sd = codes.cuda(), scale.cuda()
        _lib.check(L.wipa_embed_tokens(ptr(tok), 4, 1, 4, 0, None, ptr(cd), _lib.WIPA_FP8_E4M3, ptr(sd), ptr(pos), ptr(x), 64, sptr(s)))
    torch.cuda.synchronize()
    want = torch.stack([table[codes[r].long()] * scale[r] for r in (3, 0, 2, 1)])
    ok = ~torch.isnan(want)
    assert torch.equal(x.cpu()[ok], want[ok])
    assert ok.sum() == 254


def test_cu_limited_stream_runs_the_same_gemm(ops):
    """wipa_stream_create_cu_limited: a GEMM on a stream confined to 64 of the CUs gives the bits of the unrestricted launch;
    bad CU counts are refused."""
    import ctypes as C

    from whisper_ipa_amd import _lib
    from whisper_ipa_amd.runtime import limit_stream_cus, use_stream

    g = torch.Generator().manual_seed(5)
    A, W = torch.randn(512, 256, generator=g).cuda(), torch.randn(384, 256, generator=g).cuda()
    ref = torch.empty(512, 384, device="cuda")
    ops.gemm(A, W, ref, M=512, N=384, K=256, lda=256, ldw=256, ldc=384)
    torch.cuda.synchronize()
    limit_stream_cus(9001, 64)
    out = torch.empty_like(ref)
    with use_stream(9001) as s:
        ops.gemm(A, W, out, M=512, N=384, K=256, lda=256, ldw=256, ldc=384)
        s.synchronize()
    assert torch.equal(out, ref)
    with pytest.raises(_lib.WipaError):
        limit_stream_cus(9001, 32)  # the stream exists already
    raw = C.c_void_p()
    assert _lib.lib().wipa_stream_create_cu_limited(12, C.byref(raw)) == -1  # WIPA_ERR_ARG: not a multiple of 8
    assert _lib.lib().wipa_stream_create_cu_limited(100000, C.byref(raw)) == -1


@pytest.mark.parametrize("N", [768, 700])
@pytest.mark.parametrize("dtype,act,with_res", [(torch.float32, 0, True), (torch.float32, 1, False), (torch.bfloat16, 0, True),
                                                (torch.bfloat16, 1, False)])
def test_gemm_384x128_tiles_on_a_badly_quantised_grid(ops, dtype, act, with_res, f32_mode, N):
    """48 000 x 768 (32 clips of encoder rows, as in a fine-tune batch) is 375 tiles of 384 x 256 -- 1.46 rounds on 256 CUs --
    so the dispatcher takes 384 x 128 tiles for float32 (bf16 stays on the wide tile: no measured gain); checked on the whole
    matrix, ragged last row tile included (M = 47 990)."""
    if dtype == torch.bfloat16 and f32_mode == "split":
        pytest.skip("one bf16 run is enough")
    g = torch.Generator().manual_seed(11)
    M, K = 47990, 128  # N = 700: a ragged last column tile (6 tiles of 128, the last 60 wide)
    A = torch.randn(M, K, generator=g).to(dtype)
    W = (torch.randn(N, K, generator=g) * 0.1).to(dtype)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g) if with_res else None
    out = res.clone().cuda() if with_res else torch.empty(M, N, device="cuda")
    ops.gemm(A.cuda(), W.cuda(), out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, bias=bias.cuda(), act=act, residual=(out if with_res else None),
             f32_split=(f32_mode == "split"))
    torch.cuda.synchronize()
    y = A.double() @ W.double().T + bias.double()
    if act:
        y = torch.nn.functional.gelu(y)
    if with_res:
        y = y + res.double()
    tol = 2e-6 if (dtype == torch.float32 and f32_mode == "exact") else 1e-5 if dtype == torch.float32 else 1e-5
    assert _rel(out, y) < tol  # bf16 inputs are exact in f32 accumulation too: only the summation order differs


@pytest.mark.parametrize("Tq,Tk,causal", [(64, 1500, False), (64, 64, True), (50, 100, False), (100, 100, True), (16, 17, True)])
def test_attention_f32_mfma_forward_with_lse(Tq, Tk, causal):
    """wipa_attention in float32 with >= 16 queries runs on the f32 MFMA: output AND the saved log-sum-exp rows (what the
    backward pass consumes) against an fp64 reference; strided q / k / v views as the trainer passes them."""
    import ctypes as C

    from whisper_ipa_amd import _lib
    from whisper_ipa_amd.runtime import on_stream, ptr, sptr

    g = torch.Generator().manual_seed(Tq * 7 + Tk)
    B, H = 2, 3
    qkv = (torch.randn(B, max(Tq, Tk), 3, H, 64, generator=g) * 0.6)
    q, k, v = qkv[:, :Tq, 0], qkv[:, :Tk, 1], qkv[:, :Tk, 2]
    dev = qkv.cuda()
    qd, kd, vd = dev[:, :Tq, 0], dev[:, :Tk, 1], dev[:, :Tk, 2]
    out = torch.full((B, Tq, H, 64), 7.0, device="cuda")
    lse = torch.full((B, H, Tq), 7.0, device="cuda")
    with on_stream() as s:
        d = _lib.AttnDesc()
        d.q, d.k, d.v, d.out, d.lse = ptr(qd), ptr(kd), ptr(vd), ptr(out), ptr(lse)
        d.q_bs, d.q_rs, d.q_hs = qd.stride(0), qd.stride(1), qd.stride(2)
        d.k_bs, d.k_rs, d.k_hs = kd.stride(0), kd.stride(1), kd.stride(2)
        d.v_bs, d.v_rs, d.v_hs = vd.stride(0), vd.stride(1), vd.stride(2)
        d.o_bs, d.o_rs, d.o_hs = out.stride(0), out.stride(1), out.stride(2)
        d.B, d.H, d.Tq, d.Tk, d.causal, d.dtype = B, H, Tq, Tk, int(causal), _lib.WIPA_F32
        _lib.check(_lib.lib().wipa_attention(C.byref(d), sptr(s)), "wipa_attention")
    torch.cuda.synchronize()
    sc = torch.einsum("bqhd,bkhd->bhqk", q.double(), k.double())
    if causal:
        sc = sc + torch.triu(torch.full((Tk, Tk), float("-inf"), dtype=torch.float64), 1)[Tk - Tq:]
    ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(sc, -1), v.double())
    assert _rel(out, ref) < 2e-6
    assert (lse.cpu().double() - torch.logsumexp(sc, -1)).abs().max() < 2e-5


def test_gemm_phase_interleaved_256_tile_in_a_subprocess():
    """gemm_nt256p_kernel (WIPA_GEMM_TILE=2568, read once per process): K-tile counts 2 / 4 / 10, ragged M and N, the three
    epilogues -- against torch in float32."""
    import os
    import subprocess
    import sys

    code = r'''
import torch
from whisper_ipa_amd import ops
g = torch.Generator().manual_seed(0)
worst = 0.0
for (M, N, K, odt, act, res) in [(4096, 2560, 128, torch.bfloat16, 0, False), (4000, 2500, 256, torch.float32, 0, True),
                                 (4100, 2560, 640, torch.bfloat16, 1, False), (8192, 1280, 1280, torch.float32, 0, False)]:
    A = torch.randn(M, K, generator=g).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * 0.1).bfloat16().cuda()
    bias = torch.randn(N, generator=g).cuda()
    ldc = (N + 7) // 8 * 8
    r0 = torch.randn(M, ldc, generator=g).to(odt).cuda()
    out = r0.clone()
    ops.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldw=K, ldc=ldc, bias=bias, act=act, residual=(out if res else None))
    torch.cuda.synchronize()
    y = A.float() @ W.float().T + bias
    if act:
        y = torch.nn.functional.gelu(y)
    if res:
        y = y + r0[:, :N].float()
    err = float((out[:, :N].float() - y).abs().max() / y.abs().max())
    tol = 1e-2 if odt == torch.bfloat16 else 2e-5
    assert err < tol, (M, N, K, err)
    assert torch.equal(out[:, N:], r0[:, N:])  # padding columns untouched
    worst = max(worst, err)
print("OK", worst)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WIPA_GEMM_TILE="2568", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:] + r.stdout[-500:]


@pytest.mark.parametrize("n,slices", [(2048 * 768, 3), (1000 * 77 + 3, 4), (5, 1)])
def test_sum_slabs_ex_scale_and_residual(n, slices):
    """wipa_sum_slabs_ex: out = scale * (slabs summed in slab order) + residual, residual aliasing out allowed, tails that
    are not a multiple of 4; wipa_sum_slabs (accumulate / overwrite) on the same data."""
    from whisper_ipa_amd import _lib
    from whisper_ipa_amd.runtime import on_stream, ptr, sptr

    g = torch.Generator().manual_seed(n % 1000 + slices)
    stride = (n + 3) // 4 * 4
    slabs = torch.randn(slices, stride, generator=g).cuda()
    res = torch.randn(n, generator=g).cuda()
    ref = slabs[0, :n].clone()
    for k in range(1, slices):
        ref += slabs[k, :n]  # the kernel's order: 0, 1, 2, ... in float32
    L = _lib.lib()
    with on_stream() as s:
        out = torch.full((n,), 7.0, device="cuda")
        _lib.check(L.wipa_sum_slabs_ex(ptr(slabs), slices, stride, ptr(out), n, None, 0.5, sptr(s)), "wipa_sum_slabs_ex")
        acc = res.clone()
        _lib.check(L.wipa_sum_slabs_ex(ptr(slabs), slices, stride, ptr(acc), n, ptr(acc), 1.0, sptr(s)), "wipa_sum_slabs_ex")
        plain = torch.full((n,), 7.0, device="cuda")
        _lib.check(L.wipa_sum_slabs(ptr(slabs), slices, stride, ptr(plain), n, 0, sptr(s)), "wipa_sum_slabs")
    torch.cuda.synchronize()
    assert torch.equal(out, ref * 0.5)
    assert torch.equal(acc, res + ref)
    assert torch.equal(plain, ref)


@pytest.mark.parametrize("a_trans,w_trans", [(True, True), (True, False), (False, True)])
@pytest.mark.parametrize("M,N,K,slices", [(768, 768, 2048, 1), (300, 132, 96, 1), (1536, 768, 4096, 7), (2048, 3072, 768, 1)])
def test_gemm_k_major_operands(ops, a_trans, w_trans, M, N, K, slices, f32_mode):
    """float32 GEMM with A stored [K, M] and / or W stored [K, N] (what a linear layer's backward has: dy^T, x^T, W^T) ==
    the same product on explicitly transposed copies, bit for bit, and within f32 round-off of an fp64 reference."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) * 0.1).cuda()
    bias = torch.randn(N, generator=g).cuda()
    At, Wt = A.t().contiguous(), W.t().contiguous()  # [K, M], [K, N]
    kw = dict(M=M, N=N, K=K, ldc=N, bias=bias, f32_split=(f32_mode == "split"))
    if slices > 1:
        kw.update(k_slices=slices, slab_stride=M * N)
    shape = (slices, M, N) if slices > 1 else (M, N)
    ref = torch.empty(shape, device="cuda")
    ops.gemm(A, W, ref, lda=K, ldw=K, **kw)
    out = torch.full(shape, 7.0, device="cuda")
    ops.gemm(At if a_trans else A, Wt if w_trans else W, out, lda=(M if a_trans else K), ldw=(N if w_trans else K),
             a_trans=a_trans, w_trans=w_trans, **kw)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)  # same LDS image, same MFMA order
    full = out.sum(0) if slices > 1 else out
    exact = A.double() @ W.double().T + bias.double()
    assert _rel(full, exact) < (4e-6 if f32_mode == "exact" else 2e-5)  # contractions of up to 4096 terms


# ---------------------------------------------------------------------------------------------------------------------
# round 3: fp8 ACTIVATIONS x fp8 weights on the block-scaled fp8 matrix instruction (BASELINE.json configs[4]).
def _e4m3(t):
    from whisper_ipa_amd.whisper import dequantize_fp8_e4m3, quantize_fp8_e4m3

    codes, scale = quantize_fp8_e4m3(t)
    return codes, scale, dequantize_fp8_e4m3(codes, scale)


def test_gemm_fp8_mfma_exact_on_integers_with_asymmetric_operands(ops):
    """The operand maps of v_mfma_f32_16x16x128_f8f6f4 as the kernel uses them, checked with EXACT data: small integers are
    exact in e4m3 and their dot products are exact in f32, so C must equal the integer matrix product bit for bit.  A and W
    are asymmetric (value depends on row AND k differently), ragged M / N, several K-steps; a row <-> column swap, a wrong
    k-group or a wrong LDS chunk would all show."""
    g = torch.Generator().manual_seed(5)
    for M, N, K in ((300, 200, 256), (256, 256, 128), (513, 70, 640)):
        A = torch.randint(-3, 4, (M, K), generator=g).float()
        W = torch.randint(-3, 4, (N, K), generator=g).float()
        A[:, 0] = (torch.arange(M) % 5 - 2).float()          # row-dependent column
        W[:, 1] = (torch.arange(N) % 7 - 3).float()
        A[7, :] = (torch.arange(K) % 4).float()               # k-dependent row
        a_codes = A.to(torch.float8_e4m3fn).view(torch.uint8).cuda()
        w_codes = W.to(torch.float8_e4m3fn).view(torch.uint8).cuda()
        ones_m, ones_n = torch.ones(M, device="cuda"), torch.ones(N, device="cuda")
        out = torch.full((M, N), 9.0, device="cuda")
        ops.gemm_fp8(a_codes, ones_m, w_codes, ones_n, out)
        torch.cuda.synchronize()
        ref = A.double() @ W.double().T
        assert torch.equal(out.cpu().double(), ref), (M, N, K, (out.cpu().double() - ref).abs().max())
    assert ops.gemm_dispatch_counts(reset=True)["tile_fp8"] >= 3


@pytest.mark.parametrize("M,N,K,act,resid,out_dtype", [(1000, 768, 768, 0, True, torch.float32), (700, 1536, 768, 0, False, torch.bfloat16),
                                                      (515, 3072, 768, 1, False, torch.bfloat16), (300, 1280, 5120, 0, True, torch.float32)])
def test_gemm_fp8_mfma_with_row_scales_and_epilogues(ops, M, N, K, act, resid, out_dtype):
    """random operands quantised per row (power-of-two scales): the kernel equals the float64 product of the DEQUANTISED
    operands up to f32 accumulation, through bias / column scale / GELU / f32 residual / bf16 output."""
    g = torch.Generator().manual_seed(M + N)
    a_codes, a_scale, A = _e4m3(torch.randn(M, K, generator=g) * torch.rand(M, 1, generator=g) * 3)
    w_codes, w_scale, W = _e4m3(torch.randn(N, K, generator=g) * 0.05)
    bias = torch.randn(N, generator=g) * 0.1
    res = torch.randn(M, N, generator=g)
    out = (res.clone() if resid else torch.zeros(M, N)).to(out_dtype).cuda()
    ops.gemm_fp8(a_codes.cuda(), a_scale.cuda(), w_codes.cuda(), w_scale.cuda(), out, bias=bias.cuda(), act=act,
                 residual=out if resid else None, col_scale_n=N if not act else 0, col_scale=0.35)
    torch.cuda.synchronize()
    ref = A.double() @ W.double().T + bias.double()
    if not act:
        ref = ref * 0.35
    else:
        ref = torch.nn.functional.gelu(ref)
    if resid:
        ref = ref + res.double()
    err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < (2e-5 if out_dtype == torch.float32 else 6e-3), err


@pytest.mark.parametrize("rows,D", [(37, 768), (1000, 1280), (5, 384), (64, 5120), (9, 6144), (11, 3080)])
def test_layernorm_fp8_and_rowquant_fp8(ops, rows, D):
    """the row quantisers: codes + power-of-two scales equal the host quantiser's (Whisper.quantize_weights' rule) on the
    same values -- the LayerNorm variant up to one e4m3 ulp where its f32 rounding of LN(x) sits on a rounding boundary."""
    from whisper_ipa_amd.whisper import dequantize_fp8_e4m3, quantize_fp8_e4m3

    g = torch.Generator().manual_seed(rows + D)
    xb = (torch.randn(rows, D, generator=g) * (torch.rand(rows, 1, generator=g) * 20 + 0.01)).to(torch.bfloat16)
    xb[0] = 0  # an all-zero row
    codes, scale = ops.rowquant_fp8(xb.cuda())
    rc, rs = quantize_fp8_e4m3(xb.float())
    assert torch.equal(scale.cpu()[1:], rs[1:]) and torch.equal(codes.cpu()[1:], rc[1:])
    assert (codes.cpu()[0].view(torch.float8_e4m3fn).float() == 0).all()
    xf = torch.randn(rows, D, generator=g)
    c32, s32 = ops.rowquant_fp8(xf.cuda())
    rc, rs = quantize_fp8_e4m3(xf)
    assert torch.equal(s32.cpu(), rs) and torch.equal(c32.cpu(), rc)
    if D <= 2048:
        x = torch.randn(rows, D, generator=g) * 3 + 0.5
        w, b = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
        lc, ls = ops.layernorm_fp8(x.cuda(), w.cuda(), b.cuda())
        ref = torch.nn.functional.layer_norm(x.double(), (D,), w.double(), b.double(), 1e-5).float()
        rc, rs = quantize_fp8_e4m3(ref)
        assert torch.equal(ls.cpu(), rs)
        got, want = dequantize_fp8_e4m3(lc.cpu(), ls.cpu()), dequantize_fp8_e4m3(rc, rs)
        assert (got != want).float().mean() < 2e-3                       # boundary cases only
        assert ((got - want).abs() <= 0.126 * want.abs() + 1e-30).all()  # ... and then by one ulp (2^-3 relative)


@pytest.mark.parametrize("B,H,Tk", [(64, 12, 1500), (3, 12, 1500), (5, 6, 200), (17, 8, 95), (2, 16, 33)])
def test_cross_attention_with_absorbed_projections(B, H, Tk):
    """Round 3: decode-step cross-attention that streams the encoder output xa with the key / value projections absorbed into
    the query and the output (csrc/cross_absorbed.hip): scores_h = (q_h Wk_h) xa^T, out_h = (P_h xa) Wv_h^T + bv_h.  Against the
    float64 statement WITH explicit K = xa Wk^T and V = xa Wv^T + bv (what mlx_whisper caches), ragged frame counts, fewer than
    16 heads (the padded rows of the 16-wide MFMA dimension), strided query / output rows (the prompt prefill's layout)."""
    from whisper_ipa_amd import _lib
    from whisper_ipa_amd.runtime import on_stream, ptr, sptr

    L = _lib.lib()
    d = H * 64
    g = torch.Generator().manual_seed(B * 31 + H + Tk)
    scale = 64 ** -0.25
    q = (torch.randn(B, 2, d, generator=g) * 0.8 * scale).to(torch.bfloat16)     # rows with stride 2d: row (b, 1) is used
    wk = (torch.randn(d, d, generator=g) * 0.06).to(torch.bfloat16)
    wv = (torch.randn(d, d, generator=g) * 0.06).to(torch.bfloat16)
    bv = torch.randn(d, generator=g) * 0.1
    xa = torch.randn(B, Tk, d, generator=g).to(torch.bfloat16)
    _lib.check(L.wipa_cross_absorbed_init(d))
    with on_stream() as s:
        qd, wkT, wvd, bvd, xad = q.cuda(), wk.T.contiguous().cuda(), wv.cuda(), bv.cuda(), xa.cuda()
        out = torch.full((B, 3, d), 5.0, device="cuda", dtype=torch.bfloat16)
        nbytes = L.wipa_cross_absorbed_scratch_bytes(B, d, Tk)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        _lib.check(L.wipa_cross_absorbed_attention(ptr(qd[:, 1]), 2 * d, ptr(wkT), ptr(xad), ptr(wvd), ptr(bvd), ptr(out[:, 2]), 3 * d,
                                                   ptr(scratch), nbytes, B, H, d, Tk, scale, 0, sptr(s)), "wipa_cross_absorbed_attention")
    torch.cuda.synchronize()
    qq = q[:, 1].double().view(B, H, 64)
    K = (xa.double() @ wk.double().T * scale).view(B, Tk, H, 64)
    V = (xa.double() @ wv.double().T + bv.double()).view(B, Tk, H, 64)
    sc = torch.einsum("bhd,bthd->bht", qq, K)
    ref = torch.einsum("bht,bthd->bhd", torch.softmax(sc, -1), V).reshape(B, d)
    got = out[:, 2].float().cpu().double()
    assert (out[:, :2].float() == 5.0).all()  # rows of the other positions untouched
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 2.5e-2, err
    assert float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()) < 8e-3


@pytest.mark.parametrize("H,Tk", [(12, 1500), (6, 700)])
def test_cross_attention_absorbed_when_scores_climb_past_the_fixed_reference(H, Tk):
    """The streaming kernel takes each wave's softmax reference from the wave's FIRST 16 frames and never rescales; when later
    scores climb more than 40 above it the wave re-derives its exact maximum and streams again (csrc/cross_absorbed.hip).  Here
    the frames grow with their index along one direction of every head's absorbed query so that late scores exceed the early
    ones by ~+120 for some clips and fall by as much for others (no restart, tiny probabilities): both against float64."""
    from whisper_ipa_amd import _lib
    from whisper_ipa_amd.runtime import on_stream, ptr, sptr

    L = _lib.lib()
    B, d = 6, H * 64
    g = torch.Generator().manual_seed(H + Tk)
    scale = 64 ** -0.25
    q = (torch.randn(B, d, generator=g) * 0.8 * scale).to(torch.bfloat16)
    wk = (torch.randn(d, d, generator=g) * 0.06).to(torch.bfloat16)
    wv = (torch.randn(d, d, generator=g) * 0.06).to(torch.bfloat16)
    bv = torch.randn(d, generator=g) * 0.1
    xa = torch.randn(B, Tk, d, generator=g)
    # direction u_b = sum_h q_h Wk_h (the absorbed query summed over heads): adding ramp(t) * u_b / |u_b|^2 moves every head's score
    qk = torch.einsum("bhj,hjc->bc", q.float().view(B, H, 64), wk.float().view(H, 64, d)) * scale
    ramp = torch.linspace(0, 1, Tk).view(1, Tk, 1) * torch.tensor([120.0, -120.0, 60.0, 0.0, 200.0, 45.0]).view(B, 1, 1)
    xa = (xa + ramp * H * qk.view(B, 1, d) / qk.pow(2).sum(-1).view(B, 1, 1)).to(torch.bfloat16)
    _lib.check(L.wipa_cross_absorbed_init(d))
    with on_stream() as s:
        qd, wkT, wvd, bvd, xad = q.cuda(), wk.T.contiguous().cuda(), wv.cuda(), bv.cuda(), xa.cuda()
        out = torch.empty(B, d, device="cuda", dtype=torch.bfloat16)
        nbytes = L.wipa_cross_absorbed_scratch_bytes(B, d, Tk)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        for _ in range(2):
            _lib.check(L.wipa_cross_absorbed_attention(ptr(qd), d, ptr(wkT), ptr(xad), ptr(wvd), ptr(bvd), ptr(out), d, ptr(scratch), nbytes,
                                                       B, H, d, Tk, scale, 0, sptr(s)), "wipa_cross_absorbed_attention")
    torch.cuda.synchronize()
    # reference from the bf16-rounded absorbed queries the kernel itself uses (the ramp amplifies their rounding: a score of 200
    # moves by ~0.4 when q' moves by one bf16 ulp, which is a property of the input, not of the kernel)
    qp = (torch.einsum("bhj,hjc->bhc", q.double().view(B, H, 64), wk.double().view(H, 64, d)) * scale).to(torch.bfloat16).double()
    sc = torch.einsum("bhc,btc->bht", qp, xa.double())
    assert float((sc.amax(-1) - sc[..., :16].amax(-1)).max()) > 60  # the restart case is exercised
    px = torch.einsum("bht,btc->bhc", torch.softmax(sc, -1), xa.double())
    ref = (torch.einsum("bhc,hjc->bhj", px, wv.double().view(H, 64, d)) + bv.double().view(H, 64)).reshape(B, d)
    got = out.float().cpu().double()
    assert torch.isfinite(got).all()
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 2.5e-2, err


def test_cross_attention_absorbed_is_deterministic_with_four_streams_in_flight():
    """Regression test of two timing hazards found in round 3 (both invisible with one pass in flight): a destination
    register of ds_read_b64_tr_b16 copied before its data had arrived (the wait sat in a later asm statement), and the merge
    kernel's single-thread weight section.  The same call on four HIP streams at once, repeated: every output and every
    partial result must equal the single-stream ones bit for bit."""
    from whisper_ipa_amd import _lib
    from whisper_ipa_amd.runtime import ptr

    L = _lib.lib()
    B, H, Tk = 64, 12, 1500
    d = H * 64
    _lib.check(L.wipa_cross_absorbed_init(d))
    g = torch.Generator(device="cuda").manual_seed(0)
    xa = torch.randn(B, Tk, d, device="cuda", generator=g).bfloat16()
    q = (torch.randn(B, d, device="cuda", generator=g) * 0.3).bfloat16()
    wkT = (torch.randn(d, d, device="cuda", generator=g) * 0.05).bfloat16()
    wv = (torch.randn(d, d, device="cuda", generator=g) * 0.05).bfloat16()
    bv = torch.randn(d, device="cuda", generator=g) * 0.1
    nbytes = L.wipa_cross_absorbed_scratch_bytes(B, d, Tk)
    streams = [torch.cuda.Stream() for _ in range(4)]
    outs = [torch.empty(B, d, device="cuda", dtype=torch.bfloat16) for _ in range(4)]
    scr = [torch.zeros(nbytes, dtype=torch.uint8, device="cuda") for _ in range(4)]
    torch.cuda.synchronize()

    def call(i):
        _lib.check(L.wipa_cross_absorbed_attention(ptr(q), d, ptr(wkT), ptr(xa), ptr(wv), ptr(bv), ptr(outs[i]), d, ptr(scr[i]), nbytes, B, H, d,
                                                   Tk, 64 ** -0.25, 0, streams[i].cuda_stream))

    call(0)
    torch.cuda.synchronize()
    ref, ref_scr = outs[0].clone(), scr[0].clone()
    for _ in range(6):
        for _ in range(12):
            for i in range(4):
                call(i)
        torch.cuda.synchronize()
        for i in range(4):
            assert torch.equal(outs[i], ref), (i, int((outs[i] != ref).sum()))
            assert torch.equal(scr[i][: nbytes - 1024], ref_scr[: nbytes - 1024]), i

@pytest.mark.parametrize("B,H,n_slabs", [(64, 12, 2), (5, 6, 0), (19, 16, 4)])
def test_absorbed_cross_block_with_the_out_projection_in_its_third_launch(B, H, n_slabs):
    """Round 4: wipa_decode_cross_absorbed_block_out = the absorbed cross block whose merge launch also runs the cross-attention
    OUT projection per head (H slabs, summed by the next LayerNorm).  Against the plain block + a float64 out projection of its
    bf16 output: same x_out, and x + sum_h slab_h == x + out Wo^T + bo up to f32 summation order; wipa_add_slabs_layernorm over
    the H slabs (the 16-slab form) == LayerNorm of that sum.  Ragged batches (not a multiple of the 4-clip merge tile or of the
    16-clip prologue tile), 6 / 12 / 16 heads."""
    import ctypes as C

    from whisper_ipa_amd import _lib
    from whisper_ipa_amd.runtime import on_stream, ptr, sptr

    L = _lib.lib()
    d, Tk = H * 64, 333
    g = torch.Generator(device="cuda").manual_seed(B + H)
    rn = lambda *sh, s=1.0: torch.randn(*sh, device="cuda", generator=g) * s
    _lib.check(L.wipa_cross_absorbed_init(d))
    xa = rn(B, Tk, d).bfloat16()
    x_in, slabs_in = rn(B, d), rn(max(n_slabs, 1), B, d, s=0.3)
    ln_w, ln_b = 1 + 0.1 * rn(d), 0.1 * rn(d)
    wq, bq, wkT = rn(d, d, s=0.05).bfloat16(), rn(d, s=0.1), rn(d, d, s=0.05).bfloat16()
    wv, bv, wo, bo = rn(d, d, s=0.05).bfloat16(), rn(d, s=0.1), rn(d, d, s=0.05).bfloat16(), rn(d, s=0.1)
    nbytes = L.wipa_cross_absorbed_scratch_bytes(B, d, Tk)

    def desc(x_out, out):
        c = _lib.CrossBlockDesc()
        c.x_in, c.x_out, c.slabs, c.bias_o, c.ln_w, c.ln_b = ptr(x_in), ptr(x_out), ptr(slabs_in), None, ptr(ln_w), ptr(ln_b)
        c.wq, c.bq, c.kv, c.out = ptr(wq), ptr(bq), ptr(xa), (ptr(out) if out is not None else None)
        c.slab_stride, c.n_slabs, c.B, c.d, c.H, c.Tk, c.dtype = B * d, n_slabs, B, d, H, Tk, _lib.WIPA_BF16
        c.eps, c.qk_scale = 1e-5, 64 ** -0.25
        return c

    with on_stream() as s:
        x1, x2 = torch.empty(B, d, device="cuda"), torch.empty(B, d, device="cuda")
        out = torch.empty(B, d, device="cuda", dtype=torch.bfloat16)
        scr = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        slabs_out = torch.full((H, B, d), float("nan"), device="cuda")
        _lib.check(L.wipa_decode_cross_absorbed_block(C.byref(desc(x1, out)), ptr(wkT), ptr(wv), ptr(bv), ptr(scr), nbytes, sptr(s)))
        _lib.check(L.wipa_decode_cross_absorbed_block_out(C.byref(desc(x2, None)), ptr(wkT), ptr(wv), ptr(bv), ptr(wo), ptr(bo), ptr(slabs_out),
                                                          B * d, ptr(scr), nbytes, sptr(s)))
        mw, mb = 1 + 0.1 * rn(d), 0.1 * rn(d)
        y = torch.empty(B, d, device="cuda", dtype=torch.bfloat16)
        x3 = x2.clone()
        _lib.check(L.wipa_add_slabs_layernorm(ptr(x3), d, ptr(slabs_out), H, B * d, ptr(y), _lib.WIPA_BF16, d, ptr(mw), ptr(mb), B, d, 1e-5, sptr(s)))
    torch.cuda.synchronize()
    assert torch.equal(x1, x2) and torch.isfinite(slabs_out).all()
    want = x1.double() + out.double() @ wo.double().T + bo.double()
    got = x2.double() + slabs_out.double().sum(0)
    assert float((got - want).abs().max()) < 2e-5 * float(want.abs().max())
    # per-head slabs are what they say: slab h = v_h Wo[:, h]^T (+ bo in slab 0)
    for h in (0, H - 1):
        wh = out[:, h * 64:(h + 1) * 64].double() @ wo[:, h * 64:(h + 1) * 64].double().T + (bo.double() if h == 0 else 0)
        assert float((slabs_out[h].double() - wh).abs().max()) < 1e-5 * float(wh.abs().max() + 1)
    assert float((x3.double() - got).abs().max()) < 1e-5 * float(got.abs().max())  # the 16-slab LayerNorm left x + slabs in x
    ref_y = torch.nn.functional.layer_norm(got.float(), (d,), mw, mb, 1e-5)
    assert float((y.float() - ref_y).abs().max()) < 0.03  # bf16 output

@pytest.mark.parametrize("H,Tk", [(12, 1500), (6, 333), (16, 100)])
def test_absorbed_cross_block_frame_splits_are_a_property_of_the_call(H, Tk):
    """Round 4: wipa_cross_block_desc.cross_splits (wipa_model_cfg.dec_cross_splits) = 1..4 frame splits per clip of the streaming
    launch (2 = half-chip launches for pipelined passes).  Every count gives the float64 attention of the kernel's own bf16
    absorbed queries within the same bound as the default; the prologue (x_out) does not move; 0 means 4; the count a call uses
    is wipa_cross_absorbed_splits(want, Tk) (short inputs: >= two 32-frame tiles per split); and for a given count a clip's
    result does not depend on the batch it rides in (64 clips == the same clips 5 at a time, bit for bit)."""
    import ctypes as C

    from whisper_ipa_amd import _lib
    from whisper_ipa_amd.runtime import on_stream, ptr, sptr

    L = _lib.lib()
    B, d = 64, H * 64
    g = torch.Generator(device="cuda").manual_seed(H + Tk)
    rn = lambda *sh, s=1.0: torch.randn(*sh, device="cuda", generator=g) * s
    _lib.check(L.wipa_cross_absorbed_init(d))
    xa = rn(B, Tk, d).bfloat16()
    x_in, slabs_in = rn(B, d), rn(2, B, d, s=0.3)
    ln_w, ln_b = 1 + 0.1 * rn(d), 0.1 * rn(d)
    wq, bq, wkT = rn(d, d, s=0.05).bfloat16(), rn(d, s=0.1), rn(d, d, s=0.05).bfloat16()
    wv, bv = rn(d, d, s=0.05).bfloat16(), rn(d, s=0.1)
    tiles = (Tk + 31) // 32
    for want in range(0, 5):
        assert L.wipa_cross_absorbed_splits(want, Tk) == max(1, min(want or 4, tiles // 2))

    def run(splits, lo, hi):
        n = hi - lo
        nbytes = L.wipa_cross_absorbed_scratch_bytes(n, d, Tk)
        c = _lib.CrossBlockDesc()
        x_out = torch.empty(n, d, device="cuda")
        out = torch.empty(n, d, device="cuda", dtype=torch.bfloat16)
        scr = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        xi, sl, xs = x_in[lo:hi].contiguous(), slabs_in[:, lo:hi].contiguous(), xa[lo:hi].contiguous()
        c.x_in, c.x_out, c.slabs, c.bias_o, c.ln_w, c.ln_b = ptr(xi), ptr(x_out), ptr(sl), None, ptr(ln_w), ptr(ln_b)
        c.wq, c.bq, c.kv, c.out = ptr(wq), ptr(bq), ptr(xs), ptr(out)
        c.slab_stride, c.n_slabs, c.B, c.d, c.H, c.Tk, c.dtype = n * d, 2, n, d, H, Tk, _lib.WIPA_BF16
        c.eps, c.qk_scale, c.cross_splits = 1e-5, 64 ** -0.25, splits
        with on_stream() as s:
            _lib.check(L.wipa_decode_cross_absorbed_block(C.byref(c), ptr(wkT), ptr(wv), ptr(bv), ptr(scr), nbytes, sptr(s)))
        torch.cuda.synchronize()
        qp = scr[: n * 16 * d * 2].view(torch.bfloat16).view(n, 16, d)[:, :H]  # the absorbed queries the streaming launch read
        return x_out, out, qp.clone()

    x4, o4, qp4 = run(4, 0, B)
    x0, o0, _ = run(0, 0, B)
    assert torch.equal(x0, x4) and torch.equal(o0, o4)
    # float64 reference from the kernel's own bf16 absorbed queries: softmax(qp xa^T) xa Wv_h^T + bv
    sc = torch.einsum("bhd,btd->bht", qp4.double(), xa.double())
    pv = torch.einsum("bht,btd->bhd", torch.softmax(sc, -1), xa.double())
    ref = torch.einsum("bhd,hed->bhe", pv, wv.double().view(H, 64, d)).reshape(B, d) + bv.double()
    bound = 0.02 * float(ref.abs().max())
    assert float((o4.double() - ref).abs().max()) < bound
    for splits in (1, 2, 3):
        xs_, os_, qps = run(splits, 0, B)
        assert torch.equal(xs_, x4) and torch.equal(qps, qp4)
        assert float((os_.double() - ref).abs().max()) < bound, splits
        for lo in (0, 5, 59):  # the same clips, 5 at a time: bit-identical
            _, part, _ = run(splits, lo, lo + 5)
            assert torch.equal(part, os_[lo:lo + 5]), (splits, lo)
    c = _lib.CrossBlockDesc()
    c.cross_splits = 5
    assert L.wipa_decode_cross_absorbed_block(C.byref(c), None, None, None, None, 0, None) != 0

@pytest.mark.parametrize("B,V,d,n_init", [(64, 51865, 768, 4), (5, 51865, 384, 4), (33, 8200, 512, 1), (64, 51865, 1024, 4)])
def test_logits_projection_with_greedy_partials(B, V, d, n_init):
    """Round 4: wipa_logits_greedy = the persistent wide GEMM of the logits projection whose waves also keep max / arg-max / sum-exp
    of the filtered logits, + wipa_greedy_step_embed_partials = the step tail on those partials.  Against wipa_gemm +
    wipa_greedy_step_embed on the same inputs: the logits bit-identical (when asked for; to 1e-5 for <= 32 rows, where wipa_gemm
    runs another kernel), the chosen ids identical -- with
    DUPLICATED weight rows, so that maxima tie across waves, lanes and tiles and the lowest column must win -- the log-prob sums
    equal to 1e-6 relative, the EOT latch, the next position's embedding / LayerNorm rows and the position advance identical;
    with logits = NULL nothing is written to the logits buffer; both masks (first / always by position); ragged B and V."""
    from whisper_ipa_amd import _lib, ops
    from whisper_ipa_amd.runtime import on_stream, ptr, sptr

    L = _lib.lib()
    assert L.wipa_logits_greedy_supported(B, V, d, _lib.WIPA_BF16) == 1 and L.wipa_logits_greedy_supported(65, V, d, _lib.WIPA_BF16) == 0
    assert L.wipa_logits_greedy_supported(B, 4096, d, _lib.WIPA_BF16) == 0 and L.wipa_logits_greedy_supported(B, V, d, _lib.WIPA_F32) == 0
    g = torch.Generator(device="cuda").manual_seed(B + d)
    rn = lambda *sh, s=1.0: torch.randn(*sh, device="cuda", generator=g) * s
    ldl, n_ctx, eot = (V + 7) // 8 * 8, 448, 50257 if V > 50257 else 7
    W = rn(V, d, s=0.05).bfloat16()
    # ties: copies of the winning rows further up AND further down the vocabulary (other tiles, other waves, other lane groups)
    x = rn(B, d).bfloat16()
    with on_stream() as s:
        base = torch.empty(B, ldl, device="cuda")
        ops.gemm(x, W, base, M=B, N=V, K=d, lda=d, ldw=d, ldc=ldl)
    torch.cuda.synchronize()
    win = base[:, :V].argmax(1)
    for b in range(0, B, 3):
        for dst in (int(win[b]) + 16 * 2048 + 5, int(win[b]) - 4099, V - 1 - b):
            if 0 <= dst < V and dst != eot:
                W[dst] = W[int(win[b])]
    mask_always = torch.zeros(ldl, device="cuda")
    mask_always[torch.randperm(V, device="cuda", generator=g)[:200]] = float("-inf")
    mask_first = mask_always.clone()
    mask_first[int(win[0])] = float("-inf")  # the first-position mask removes row 0's winner
    emb, pos_emb = W, rn(n_ctx, d, s=0.02)
    ln_w, ln_b = 1 + 0.1 * rn(d), 0.1 * rn(d)

    def run(fused, p0, store):
        tokens = torch.full((B, n_ctx + 8), 1, dtype=torch.int32, device="cuda")
        tokens[2 % B, p0] = eot  # this row is latched
        pos = torch.tensor([p0], dtype=torch.int32, device="cuda")
        posd = torch.zeros(1, dtype=torch.int64, device="cuda")
        done = torch.zeros(1, dtype=torch.int32, device="cuda")
        slp = torch.zeros(B, device="cuda")
        nd = torch.zeros(1, dtype=torch.int32, device="cuda")
        logits = torch.full((B, ldl), 7.0, device="cuda")
        xo, yo = torch.empty(B, d, device="cuda"), torch.empty(B, d, device="cuda", dtype=torch.bfloat16)
        with on_stream() as s:
            if fused:
                nb = L.wipa_logits_greedy_partials_bytes(B)
                part = torch.empty(nb, dtype=torch.uint8, device="cuda")
                _lib.check(L.wipa_logits_greedy(ptr(x), d, ptr(W), d, ptr(logits) if store else None, ldl, B, V, d, ptr(mask_first), ptr(mask_always),
                                                ptr(pos), n_init, ptr(part), nb, sptr(s)), "wipa_logits_greedy")
                _lib.check(L.wipa_greedy_step_embed_partials(ptr(part), _lib.GREEDY_PARTS, B, ptr(tokens), n_ctx + 8, ptr(pos), ptr(posd), ptr(done),
                                                             n_init, eot, ptr(slp), ptr(nd), ptr(emb), _lib.WIPA_BF16, None, ptr(pos_emb), n_ctx,
                                                             ptr(xo), ptr(ln_w), ptr(ln_b), ptr(yo), _lib.WIPA_BF16, d, 1e-5, sptr(s)))
            else:
                ops.gemm(x, W, logits, M=B, N=V, K=d, lda=d, ldw=d, ldc=ldl)
                _lib.check(L.wipa_greedy_step_embed(ptr(logits), ldl, B, V, ptr(mask_first), ptr(mask_always), ptr(tokens), n_ctx + 8, ptr(pos),
                                                    ptr(posd), ptr(done), n_init, eot, ptr(slp), ptr(nd), ptr(emb), _lib.WIPA_BF16, None,
                                                    ptr(pos_emb), n_ctx, ptr(xo), ptr(ln_w), ptr(ln_b), ptr(yo), _lib.WIPA_BF16, d, 1e-5, sptr(s)))
        torch.cuda.synchronize()
        return tokens[:, p0 + 1].cpu(), slp.cpu(), logits.cpu(), xo.cpu(), yo.cpu(), int(pos), int(posd), int(nd), int(done)

    for p0 in (n_init - 1, n_init + 3):  # the first generated position (mask_first) and a later one (mask_always)
        ref = run(False, p0, True)
        for store in (True, False):
            got = run(True, p0, store)
            assert torch.equal(got[0], ref[0]), (p0, store, (got[0] != ref[0]).nonzero().flatten().tolist())
            assert torch.allclose(got[1], ref[1], rtol=1e-6, atol=1e-6), (p0, store)
            if store and B > 32:  # the same kernel arithmetic (wipa_gemm sends > 32 rows to the persistent wide kernel too)
                assert torch.equal(got[2][:, :V], ref[2][:, :V])
            elif store:  # <= 32 rows: wipa_gemm splits K over the waves of a workgroup -- another summation order
                assert torch.allclose(got[2][:, :V], ref[2][:, :V], rtol=1e-5, atol=1e-5)
            else:
                assert bool((got[2] == 7.0).all())  # untouched
            assert torch.equal(got[3], ref[3]) and torch.equal(got[4], ref[4]) and got[5:] == ref[5:], (p0, store)
        filt = ref[2][:, :V] + (mask_first if p0 + 1 == n_init else mask_always)[:V].cpu()
        want = filt.argmax(1).int()  # torch: first maximum
        want[2 % B] = eot
        assert torch.equal(ref[0], want) and ref[5] == p0 + 1 and ref[6] == (p0 + 1) * d and ref[8] == 0
        lp = torch.log_softmax(filt.double(), 1).gather(1, filt.argmax(1, keepdim=True)).flatten().float()
        lp[2 % B] = 0.0
        assert torch.allclose(got[1], lp, rtol=1e-4, atol=1e-4)


def test_gemm_fp8_384x256_tile_in_a_subprocess():
    """Round 4: gemm_fp8_384_kernel (fp8 x fp8 on the 384 x 256 tile; WIPA_GEMM_FP8_TILE=384 forces it, read once per process).
    EXACT on small integers with asymmetric operands (a row <-> column swap, a wrong k-group or LDS chunk would show), ragged
    M / N around the 384-row and 256-column tile edges, several K-steps; then random operands with per-row scales through bias /
    column scale / GELU / f32 residual / bf16 output against float64 on the dequantised operands; and the same calls give the
    256 x 256 kernel's bits when the products are exact."""
    import os
    import subprocess
    import sys

    code = r'''
import torch
from whisper_ipa_amd import ops
g = torch.Generator().manual_seed(5)
for M, N, K in ((800, 300, 256), (768, 512, 128), (1153, 72, 640), (900, 260, 384)):  # N % 4 == 0: the staged epilogue the tile needs
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    W = torch.randint(-3, 4, (N, K), generator=g).float()
    A[:, 0] = (torch.arange(M) % 5 - 2).float()
    W[:, 1] = (torch.arange(N) % 7 - 3).float()
    A[7, :] = (torch.arange(K) % 4).float()
    a_codes = A.to(torch.float8_e4m3fn).view(torch.uint8).cuda()
    w_codes = W.to(torch.float8_e4m3fn).view(torch.uint8).cuda()
    out = torch.full((M, N), 9.0, device="cuda")
    ops.gemm_fp8(a_codes, torch.ones(M, device="cuda"), w_codes, torch.ones(N, device="cuda"), out)
    torch.cuda.synchronize()
    ref = A.double() @ W.double().T
    assert torch.equal(out.cpu().double(), ref), (M, N, K, float((out.cpu().double() - ref).abs().max()))
counts = ops.gemm_dispatch_counts(reset=True)
assert counts["tile_fp8_384"] == 4 and counts["tile_fp8"] == 4, counts

def e4m3(x):
    from whisper_ipa_amd.whisper import quantize_fp8_e4m3, dequantize_fp8_e4m3
    c, s = quantize_fp8_e4m3(x)
    return c, s, dequantize_fp8_e4m3(c, s)

worst = 0.0
for (M, N, K, act, resid, odt) in [(1000, 768, 768, 0, True, torch.float32), (900, 1536, 768, 0, False, torch.bfloat16),
                                   (1200, 3072, 768, 1, False, torch.bfloat16), (770, 1280, 5120, 0, True, torch.float32)]:
    g = torch.Generator().manual_seed(M + N)
    a_codes, a_scale, A = e4m3(torch.randn(M, K, generator=g) * torch.rand(M, 1, generator=g) * 3)
    w_codes, w_scale, W = e4m3(torch.randn(N, K, generator=g) * 0.05)
    bias = torch.randn(N, generator=g) * 0.1
    res = torch.randn(M, N, generator=g)
    out = (res.clone() if resid else torch.zeros(M, N)).to(odt).cuda()
    ops.gemm_fp8(a_codes.cuda(), a_scale.cuda(), w_codes.cuda(), w_scale.cuda(), out, bias=bias.cuda(), act=act,
                 residual=out if resid else None, col_scale_n=N if not act else 0, col_scale=0.35)
    torch.cuda.synchronize()
    ref = A.double() @ W.double().T + bias.double()
    ref = torch.nn.functional.gelu(ref) if act else ref * 0.35
    if resid:
        ref = ref + res.double()
    err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < (2e-5 if odt == torch.float32 else 6e-3), (M, N, K, err)
    worst = max(worst, err)
assert ops.gemm_dispatch_counts(reset=True)["tile_fp8_384"] == 4
print("OK", worst)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WIPA_GEMM_FP8_TILE="384", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:] + r.stdout[-500:]


def test_library_streams_are_the_librarys_own_and_created_together():
    """runtime.stream(): the library's own HIP streams (wipa_stream_create), the first OWN_STREAM_COUNT created TOGETHER at the
    first request -- ROCm hands hardware queues to streams in creation order, so the passes in flight of pipeline.py sit on
    distinct queues even with the default four (bench: 71.8 ms per pass at GPU_MAX_HW_QUEUES=4 against 85.9 with
    torch's pool streams).  Distinct handles, usable, stable across calls; ids beyond the eager set are created on demand."""
    from whisper_ipa_amd import runtime as RT

    s0 = RT.stream(0)
    own = RT._own_streams[torch.cuda.current_device()]
    assert len(own) == RT.OWN_STREAM_COUNT == 8 and own[0] is s0
    handles = {s.cuda_stream for s in own}
    assert len(handles) == 8 and 0 not in handles  # eight different HIP streams, none of them the null stream
    assert all(isinstance(s, torch.cuda.ExternalStream) for s in own)
    assert RT.stream(3) is own[3] and RT.stream(3) is RT.stream(3)
    far = RT.stream(4242)
    assert far.cuda_stream not in handles and RT.stream(4242) is far
    x = torch.arange(1024, device="cuda", dtype=torch.float32)
    outs = []
    for i in range(8):
        with RT.use_stream(i) as s:
            s.wait_stream(torch.cuda.current_stream())
            outs.append((x * (i + 1)).sum())
    torch.cuda.synchronize()
    assert [float(o) for o in outs] == [float(x.sum()) * (i + 1) for i in range(8)]
