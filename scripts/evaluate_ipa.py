"""IPA metrics with the surface the training loop needs from the reference's scripts/evaluate_ipa.py:
``tokenize_ipa`` (:27-65), ``normalize_ipa_for_comparison`` (:68-77), ``phone_error_rate`` (:80-105),
``evaluate_batch`` (:346-378) -> {'per','pfer','per_std','pfer_std','num_samples'}.

Host-side string work (O(len^2) DP), not GPU work.  The reference segments with panphon and scores
PFER with panphon's 24 articulatory features; panphon and its feature table are not in this image,
so segmentation is the reference's own Unicode fallback rule (base character + combining marks and
spacing modifier letters U+02B0-U+02FF, :51-64) -- it satisfies all nine tokenisation assertions of
the reference (:449-457) -- and PFER uses panphon when it can be imported, otherwise it falls back
to PER and says so in the result (``pfer_is_per_fallback``).
"""
from __future__ import annotations

import unicodedata
from typing import Dict, List, Sequence

import numpy as np


def tokenize_ipa(text: str) -> List[str]:
    text = text.replace(" ", "")
    segments: List[str] = []
    for ch in text:
        cat = unicodedata.category(ch)
        is_modifier = cat.startswith("M") or (cat == "Lm" and "ʰ" <= ch <= "˿")
        if segments and is_modifier:
            segments[-1] += ch
        else:
            segments.append(ch)
    return segments


def normalize_ipa_for_comparison(text: str) -> str:
    text = unicodedata.normalize("NFC", text).replace(" ", "")
    return text.replace("g", "ɡ")  # Latin g -> IPA g


def edit_distance(a: Sequence, b: Sequence) -> int:
    """Levenshtein distance (substitution, insertion, deletion all cost 1)."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def phone_error_rate(reference: str, hypothesis: str) -> float:
    ref, hyp = tokenize_ipa(reference), tokenize_ipa(hypothesis)
    if not ref:
        return 0.0 if not hyp else 100.0
    return edit_distance(ref, hyp) / len(ref) * 100.0


def _panphon_table():
    try:
        import panphon  # noqa: F401

        return panphon.FeatureTable()
    except Exception:
        return None


def phone_feature_error_rate(reference: str, hypothesis: str, ft=None) -> float:
    """Hamming feature edit distance / 24 per reference phone (reference :108-213) when panphon is
    available; PER otherwise."""
    ft = ft or _panphon_table()
    if ft is None:
        return phone_error_rate(reference, hypothesis)
    ref, hyp = tokenize_ipa(reference), tokenize_ipa(hypothesis)
    if not ref:
        return 0.0 if not hyp else 100.0

    def vec(p):
        v = ft.word_to_vector_list(p, numeric=True)
        return np.asarray(v[0] if v else [0] * 24, dtype=np.float64)

    R, H = [vec(p) for p in ref], [vec(p) for p in hyp]
    n, m = len(R), len(H)
    D = np.zeros((n + 1, m + 1))
    D[:, 0] = np.arange(n + 1)
    D[0, :] = np.arange(m + 1)
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            sub = np.abs(R[i - 1] - H[j - 1]).sum() / 2.0 / 24.0
            D[i, j] = min(D[i - 1, j] + 1, D[i, j - 1] + 1, D[i - 1, j - 1] + sub)
    return D[n, m] / n * 100.0


def evaluate_batch(references: List[str], hypotheses: List[str]) -> Dict:
    ft = _panphon_table()
    pers, pfers = [], []
    for r, h in zip(references, hypotheses):
        r, h = normalize_ipa_for_comparison(r), normalize_ipa_for_comparison(h)
        pers.append(phone_error_rate(r, h))
        pfers.append(phone_feature_error_rate(r, h, ft))
    if not pers:
        return {"per": 0.0, "pfer": 0.0, "per_std": 0.0, "pfer_std": 0.0, "num_samples": 0, "pfer_is_per_fallback": ft is None}
    return {"per": float(np.mean(pers)), "pfer": float(np.mean(pfers)), "per_std": float(np.std(pers)),
            "pfer_std": float(np.std(pfers)), "num_samples": len(pers), "pfer_is_per_fallback": ft is None}
