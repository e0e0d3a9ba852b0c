"""CPU: the C-ABI library loads and exports every symbol include/wipa.h declares, the ctypes
structs mirror the C layouts, and importing the package needs no GPU.  No compute calls."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "wipa.h")


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g

    g.build()
    from whisper_ipa_amd import _lib

    return _lib


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wipa_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound(built):
    names = _declared()
    assert len(names) >= 20
    lib = built.lib()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in wipa.h but not exported by libwipa.so"
        assert n in built.SIGNATURES, f"{n} has no ctypes signature in _lib.py"
    extra = set(built.SIGNATURES) - set(names)
    assert not extra, f"bound but not declared in wipa.h: {extra}"


def test_version_and_error_string(built):
    lib = built.lib()
    assert lib.wipa_version() >= 100
    assert isinstance(lib.wipa_last_error(), bytes)


def test_struct_layouts_match_c(built, tmp_path):
    """Compile a tiny C program against wipa.h and compare sizeof/offsetof with ctypes."""
    prog = tmp_path / "lay.c"
    prog.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "wipa.h"\n'
        "int main(){\n"
        'printf("%zu %zu %zu %zu\\n", sizeof(wipa_gemm_desc), sizeof(wipa_attn_desc), sizeof(wipa_model_cfg), sizeof(wipa_dec_layout));\n'
        'printf("%zu %zu %zu %zu\\n", offsetof(wipa_gemm_desc, lda), offsetof(wipa_gemm_desc, M), offsetof(wipa_gemm_desc, col_scale), offsetof(wipa_gemm_desc, cg_in));\n'
        'printf("%zu %zu %zu\\n", offsetof(wipa_attn_desc, q_bs), offsetof(wipa_attn_desc, B), offsetof(wipa_attn_desc, dtype));\n'
        'printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(wipa_self_block_desc), sizeof(wipa_cross_block_desc), offsetof(wipa_self_block_desc, B), '
        'offsetof(wipa_self_block_desc, qk_scale), offsetof(wipa_cross_block_desc, n_slabs), offsetof(wipa_cross_block_desc, qk_scale), '
        'offsetof(wipa_gemm_desc, ln_eps));\n'
        "return 0;}\n"
    )
    exe = tmp_path / "lay"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split("\n")
    sizes = [int(x) for x in out[0].split()]
    assert sizes == [C.sizeof(built.GemmDesc), C.sizeof(built.AttnDesc), C.sizeof(built.ModelCfg), C.sizeof(built.DecLayout)]
    g = [int(x) for x in out[1].split()]
    assert g == [built.GemmDesc.lda.offset, built.GemmDesc.M.offset, built.GemmDesc.col_scale.offset, built.GemmDesc.cg_in.offset]
    a = [int(x) for x in out[2].split()]
    assert a == [built.AttnDesc.q_bs.offset, built.AttnDesc.B.offset, built.AttnDesc.dtype.offset]
    f = [int(x) for x in out[3].split()]
    assert f == [C.sizeof(built.SelfBlockDesc), C.sizeof(built.CrossBlockDesc), built.SelfBlockDesc.B.offset,
                 built.SelfBlockDesc.qk_scale.offset, built.CrossBlockDesc.n_slabs.offset, built.CrossBlockDesc.qk_scale.offset,
                 built.GemmDesc.ln_eps.offset]


def test_missing_library_fails_loudly(monkeypatch, built):
    """No silent fallback: a missing .so is an error, not a CPU path."""
    from whisper_ipa_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libwipa.so")
    with pytest.raises(_lib.WipaError):
        _lib.lib()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "whisper_ipa_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.replace("# oracle", ""), f"{f} mentions the oracle"


def test_decoder_entry_points_refuse_a_state_blob_smaller_than_the_layout(built):
    """wipa_decoder_begin / run / prefill / set_audio take the caller's blob size and return an error BEFORE any launch when it
    is smaller than wipa_decoder_layout(cfg, B).total_bytes -- the layout depends on the configuration (cached vs absorbed
    cross-attention: 24 x at whisper-small), so a blob must never outlive a change of it.  No GPU is touched on this path."""
    lib = built.lib()
    small = dict(n_mels=80, n_audio_ctx=1500, n_audio_state=768, n_audio_head=12, n_audio_layer=12, n_vocab=51865, n_text_ctx=448,
                 n_text_state=768, n_text_head=12, n_text_layer=12, dtype=built.WIPA_BF16)
    cached, absorbed = built.ModelCfg(**small, dec_cross_absorbed=0), built.ModelCfg(**small, dec_cross_absorbed=1)
    lc, la = built.DecLayout(), built.DecLayout()
    assert lib.wipa_decoder_layout(C.byref(cached), 64, C.byref(lc)) == 0 and lib.wipa_decoder_layout(C.byref(absorbed), 64, C.byref(la)) == 0
    assert lc.total_bytes > 1.5 * la.total_bytes  # 3.5 GB of K / V against 0.15 GB of features
    fake = C.c_void_p(0x1000)  # never dereferenced: the size check comes first
    init = (C.c_int32 * 4)(1, 2, 3, 4)
    tab = (C.c_void_p * 4)()
    rc = lib.wipa_decoder_begin(C.byref(cached), fake, la.total_bytes, 64, init, 4, None)
    assert rc != 0 and b"state blob" in lib.wipa_last_error()
    rc = lib.wipa_decoder_run(C.byref(cached), tab, fake, la.total_bytes, 64, 4, 50257, fake, fake, 1, 0, None)
    assert rc != 0 and b"state blob" in lib.wipa_last_error()
    rc = lib.wipa_decoder_prefill(C.byref(cached), tab, fake, la.total_bytes, 64, 4, 50257, fake, fake, 0, None)
    assert rc != 0 and b"state blob" in lib.wipa_last_error()
    rc = lib.wipa_decoder_set_audio(C.byref(cached), tab, fake, fake, la.total_bytes, 64, None)
    assert rc != 0 and b"state blob" in lib.wipa_last_error()
