"""Whisper tokenizer with the surface of ``mlx_whisper.tokenizer`` the reference uses
(scripts/ipa_data_loader.py:146-152,113-120; scripts/train_whisper_ipa.py:242,372):
``get_tokenizer(multilingual=True)`` -> object with ``.encode``, ``.decode``, ``.eot``,
``.sot``, ``.sot_sequence_including_notimestamps``, ``.language`` (settable),
``.non_speech_tokens``.

The byte-level BPE is implemented here (host side, pure Python).  The rank table
(``multilingual.tiktoken``: base64 token -> rank per line, 50 257 ranks) is an asset of
mlx_whisper that is NOT in the reference tree nor in this image; pass its path as
``vocab_path`` or set ``WIPA_TIKTOKEN``.  Without it the tokenizer runs in *byte-fallback*
mode: the 256 byte-level ranks of the GPT-2 vocabulary family (which are Whisper's ids for every
character it has no merge for -- most IPA symbols) and no merges: framing, special ids, round trips
and the unmerged known answers of the reference hold; merged ids ('ə' -> [7250]) do not.
"""
from __future__ import annotations

import base64
import os
import re
from dataclasses import dataclass, field
from functools import lru_cache
from typing import Dict, List, Optional, Sequence, Tuple

LANGUAGES = (
    "en zh de es ru ko fr ja pt tr pl ca nl ar sv it id hi fi vi he uk el ms cs ro da hu ta no th ur hr bg lt la mi ml cy "
    "sk te fa lv bn sr az sl kn et mk br eu is hy ne mn bs kk sq sw gl mr pa si km sn yo so af oc ka be tg sd gu am yi lo "
    "uz fo ht ps tk nn mt sa lb my bo tl mg as tt haw ln ha ba jw su yue"
).split()

# tokenizer.non_speech_tokens for the multilingual (<= large-v2) vocabulary: published ids.
NON_SPEECH_TOKENS_MULTI = (
    1, 2, 7, 8, 9, 10, 14, 25, 26, 27, 28, 29, 31, 58, 59, 60, 61, 62, 63, 90, 91, 92, 93, 359, 503, 522, 542, 873,
    893, 902, 918, 922, 931, 1350, 1853, 1982, 2460, 2627, 3246, 3253, 3268, 3536, 3846, 3961, 4183, 4667, 6585, 6647,
    7273, 9061, 9383, 10428, 10929, 11938, 12033, 12331, 12562, 13793, 14157, 14635, 15265, 15618, 16553, 16604, 18362,
    18956, 20075, 21675, 22520, 26130, 26161, 26435, 28279, 29464, 31650, 32302, 32470, 36865, 42863, 47425, 49870,
    50254,
)

# GPT-2 pre-tokeniser (tiktoken "gpt2" pat_str), written for the stdlib-compatible `regex` module if
# present, else an ASCII-class approximation that is exact for the IPA / Latin strings in the datasets'
# pre-split units (letters, marks and digits are classified with unicodedata below).
_PAT = r"""'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""


def _compile_pat():
    try:
        import regex

        return regex.compile(_PAT)
    except Exception:  # pragma: no cover
        return re.compile(r"'s|'t|'re|'ve|'m|'ll|'d| ?[^\W\d_]+| ?\d+| ?[^\s\w]+|\s+(?!\S)|\s+", re.UNICODE)


def load_tiktoken_ranks(path: str) -> Dict[bytes, int]:
    ranks = {}
    with open(path, "rb") as f:
        for line in f:
            if line.strip():
                tok, rank = line.split()
                ranks[base64.b64decode(tok)] = int(rank)
    return ranks


def _bpe(ranks: Dict[bytes, int], piece: bytes) -> List[int]:
    """Byte-pair merge by lowest rank (tiktoken's algorithm), pure Python: the statement the native
    wipa_bpe_encode_piece (csrc/text_host.cpp) is tested against; the Tokenizer itself calls the native one."""
    if piece in ranks:
        return [ranks[piece]]
    parts = [bytes([b]) for b in piece]
    while len(parts) > 1:
        best, best_rank = -1, None
        for i in range(len(parts) - 1):
            r = ranks.get(parts[i] + parts[i + 1])
            if r is not None and (best_rank is None or r < best_rank):
                best, best_rank = i, r
        if best_rank is None:
            break
        parts[best : best + 2] = [parts[best] + parts[best + 1]]
    return [ranks[p] for p in parts]


class _NativeBPE:
    """The merge table inside libwipa (wipa_bpe_*): built once per Tokenizer, freed with it."""

    def __init__(self, ranks: Dict[bytes, int]):
        import ctypes as C

        from . import _lib

        self._C, self._L = C, _lib.lib()
        toks = list(ranks.items())
        blob = b"".join(t for t, _ in toks)
        lens = (C.c_int32 * len(toks))(*[len(t) for t, _ in toks])
        rk = (C.c_int32 * len(toks))(*[r for _, r in toks])
        self._h = self._L.wipa_bpe_create(blob, lens, rk, len(toks))
        if not self._h:
            raise _lib.WipaError("wipa_bpe_create failed: " + (self._L.wipa_last_error() or b"").decode())

    def encode_piece(self, piece: bytes) -> List[int]:
        C = self._C
        buf = (C.c_int32 * max(len(piece), 1))()
        n = self._L.wipa_bpe_encode_piece(self._h, piece, len(piece), buf, len(buf))
        if n < 0:
            from . import _lib

            raise _lib.WipaError(f"wipa_bpe_encode_piece failed ({n}): " + (self._L.wipa_last_error() or b"").decode())
        return list(buf[:n])

    def decode_bytes(self, ids: Sequence[int]) -> Optional[bytes]:
        """bytes of ids that are all in the table, else None"""
        C = self._C
        arr = (C.c_int32 * len(ids))(*ids)
        cap = 64 + 16 * len(ids)
        while True:
            buf = (C.c_uint8 * cap)()
            n = self._L.wipa_bpe_decode(self._h, arr, len(ids), buf, cap)
            if n >= 0:
                return bytes(buf[:n])
            if n == -1 and cap < (1 << 24):  # WIPA_ERR_ARG: buffer too small
                cap *= 8
                continue
            return None

    def __del__(self):
        try:
            if self._h:
                self._L.wipa_bpe_free(self._h)
                self._h = None
        except Exception:
            pass


@dataclass
class Tokenizer:
    ranks: Dict[bytes, int]
    num_languages: int = 99
    language: Optional[str] = "en"
    task: Optional[str] = "transcribe"
    byte_fallback: bool = False
    special_tokens: Dict[str, int] = field(default_factory=dict)
    sot_sequence: Tuple[int, ...] = ()

    def __post_init__(self):
        n = len(self.ranks)
        specials = ["<|endoftext|>", "<|startoftranscript|>"] + [f"<|{l}|>" for l in LANGUAGES[: self.num_languages]] + [
            "<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>", "<|nospeech|>", "<|notimestamps|>"
        ] + [f"<|{i * 0.02:.2f}|>" for i in range(1501)]
        base = n if not self.byte_fallback else 50257
        self.special_tokens = {s: base + i for i, s in enumerate(specials)}
        self._decoder = {v: k for k, v in self.ranks.items()}
        self._special_decoder = {v: k for k, v in self.special_tokens.items()}
        self._pat = _compile_pat()
        self._native = _NativeBPE(self.ranks)
        # like mlx_whisper, sot_sequence is frozen here: assigning .language later (ipa_data_loader.py:152)
        # does NOT change it.
        seq = [self.sot]
        if self.language is not None:
            seq.append(self.sot + 1 + LANGUAGES.index(self.language))
        if self.task is not None:
            seq.append(self.transcribe if self.task == "transcribe" else self.translate)
        self.sot_sequence = tuple(seq)

    # ---- special ids
    @property
    def eot(self) -> int: return self.special_tokens["<|endoftext|>"]
    @property
    def sot(self) -> int: return self.special_tokens["<|startoftranscript|>"]
    @property
    def translate(self) -> int: return self.special_tokens["<|translate|>"]
    @property
    def transcribe(self) -> int: return self.special_tokens["<|transcribe|>"]
    @property
    def sot_lm(self) -> int: return self.special_tokens["<|startoflm|>"]
    @property
    def sot_prev(self) -> int: return self.special_tokens["<|startofprev|>"]
    @property
    def no_speech(self) -> int: return self.special_tokens["<|nospeech|>"]
    @property
    def no_timestamps(self) -> int: return self.special_tokens["<|notimestamps|>"]
    @property
    def timestamp_begin(self) -> int: return self.special_tokens["<|0.00|>"]
    @property
    def sot_sequence_including_notimestamps(self) -> Tuple[int, ...]:
        return tuple(list(self.sot_sequence) + [self.no_timestamps])
    @property
    def all_language_tokens(self) -> Tuple[int, ...]:
        return tuple(self.sot + 1 + i for i in range(self.num_languages))
    @property
    def n_vocab(self) -> int:
        return max(self.special_tokens.values()) + 1

    def to_language_token(self, language: str) -> int:
        return self.special_tokens[f"<|{language}|>"]

    # ---- text <-> ids
    def encode(self, text: str) -> List[int]:
        out: List[int] = []
        for piece in self._pat.findall(text):
            out.extend(self._native.encode_piece(piece.encode("utf-8")))
        return out

    def decode(self, token_ids: Sequence[int]) -> str:
        """Like tiktoken decode on the ids below timestamp_begin; special tokens render as <|...|>
        (train_whisper_ipa.py:372-379 strips them with a regex)."""
        buf, out = bytearray(), []
        for t in token_ids:
            t = int(t)
            if t in self._decoder:
                buf.extend(self._decoder[t])
            else:
                if buf:
                    out.append(buf.decode("utf-8", errors="replace"))
                    buf = bytearray()
                if t in self._special_decoder:
                    if t < self.timestamp_begin:
                        out.append(self._special_decoder[t])
                else:
                    out.append(f"<|id{t}|>")
        if buf:
            out.append(buf.decode("utf-8", errors="replace"))
        return "".join(out)

    @property
    def non_speech_tokens(self) -> Tuple[int, ...]:
        """mlx_whisper.tokenizer.Tokenizer.non_speech_tokens: derived from the vocabulary when it
        is loaded, else the published id list of the multilingual vocabulary."""
        if self.byte_fallback:
            return NON_SPEECH_TOKENS_MULTI
        symbols = list('"#()*+/:;<=>@[\\]^_`{|}~「」『』')
        symbols += "<< >> <<< >>> -- --- -( -[ (' (\" (( )) ((( ))) [[ ]] {{ }} ♪♪ ♪♪♪".split()
        miscellaneous = set("♩♪♫♬♭♮♯")
        result = {self.encode(" -")[0], self.encode(" '")[0]}
        for symbol in symbols + list(miscellaneous):
            for toks in (self.encode(symbol), self.encode(" " + symbol)):
                if len(toks) == 1 or symbol in miscellaneous:
                    result.add(toks[0])
        return tuple(sorted(result))


@lru_cache(maxsize=8)
def get_tokenizer(multilingual: bool = True, *, num_languages: int = 99, language: Optional[str] = None,
                  task: Optional[str] = None, vocab_path: Optional[str] = None) -> Tokenizer:
    if not multilingual:
        raise NotImplementedError("the reference only uses the multilingual tokenizer (ipa_data_loader.py:149)")
    language = language or "en"
    task = task or "transcribe"
    path = vocab_path or os.environ.get("WIPA_TIKTOKEN")
    if path:
        if not os.path.exists(path):
            raise FileNotFoundError(f"Whisper vocabulary {path!r} (vocab_path / WIPA_TIKTOKEN) does not exist")
        return Tokenizer(load_tiktoken_ranks(path), num_languages, language, task, byte_fallback=False)
    _warn_byte_fallback()
    return Tokenizer(gpt2_byte_ranks(), num_languages, language, task, byte_fallback=True)


def gpt2_byte_ranks() -> Dict[bytes, int]:
    """Ranks 0..255 of the byte-level BPE vocabularies of the GPT-2 family (Whisper's multilingual.tiktoken included): the
    printable non-space Latin-1 bytes first (33..126, 161..172, 174..255 -> 0..187), then the remaining bytes in order
    (-> 188..255).  These ARE Whisper's ids for every character the vocabulary has no merge for -- which is most of IPA:
    the reference's known answers 'p' -> [79], 'ɪ' -> [133, 103], 'tʰ' -> [83, 134, 108], 'n̩' -> [77, 136, 102]
    (WHISPER_IPA_RESEARCH_STANDALONE.md:280-305,498-505) and the blank token ' ' -> 220 follow from this table alone."""
    order = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    order += [b for b in range(256) if b not in set(order)]
    return {bytes([b]): i for i, b in enumerate(order)}


_warned = False


def _warn_byte_fallback() -> None:
    global _warned
    if not _warned:
        _warned = True
        import sys

        print("WARNING: no Whisper vocabulary (set WIPA_TIKTOKEN to mlx_whisper's assets/multilingual.tiktoken): the tokenizer "
              "runs in BYTE-FALLBACK mode -- text is tokenised byte by byte (Whisper's byte-level ids, no merges: e.g. 'ə' -> "
              "[133, 247] instead of [7250]); special ids, framing and the decode loop are unaffected.  Fine for synthetic "
              "weights, WRONG for a pretrained checkpoint.", file=sys.stderr, flush=True)


class VocabularyError(RuntimeError):
    pass


def require_real_vocabulary(tok: Tokenizer, allow_byte_fallback: bool = False, what: str = "this run") -> Tokenizer:
    """Entry points that pair the tokenizer with REAL weights (fine-tuning, evaluation, transcription) call this: a
    pretrained Whisper must not be trained or scored on byte-fallback ids by accident.  ``allow_byte_fallback`` (the
    scripts' --allow-byte-fallback, or WIPA_ALLOW_BYTE_FALLBACK=1) is the explicit opt-in for synthetic-weight runs."""
    if tok.byte_fallback and not (allow_byte_fallback or os.environ.get("WIPA_ALLOW_BYTE_FALLBACK") == "1"):
        raise VocabularyError(
            f"{what}: the Whisper vocabulary is missing (WIPA_TIKTOKEN is unset), so text would be tokenised as raw bytes, not "
            "Whisper BPE ids.  Point WIPA_TIKTOKEN at multilingual.tiktoken, or pass --allow-byte-fallback for a run on "
            "synthetic weights.")
    return tok
