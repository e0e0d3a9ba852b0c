"""IPA metrics with the surface of the reference's scripts/evaluate_ipa.py: ``tokenize_ipa`` (:27-65),
``normalize_ipa_for_comparison`` (:68-77), ``phone_error_rate`` (:80-105), ``PFERCalculator`` (Hamming, :108-213),
``PFERCalculatorCosine`` (:216-287), ``phone_feature_error_rate[_cosine]`` (:300-343), ``evaluate_batch`` (:346-378)
-> {'per','pfer','per_std','pfer_std','num_samples','per_scores','pfer_scores'}.

Host-side string work (O(len^2) DP), not GPU work.  The reference takes the 24 articulatory features and its primary
segmentation from ``panphon.FeatureTable``; panphon is not in this image.  The feature source is therefore pluggable:

* ``panphon`` when it imports;
* else a panphon-format feature CSV named by ``WIPA_PANPHON_CSV`` (panphon's ``data/ipa_all.csv``: header
  ``ipa,syl,son,...``, values ``+ - 0``) through :class:`CsvFeatureTable`: an exact row when the segment has one, else the
  row of its BASE character (panphon itself would apply its diacritic rules to that base vector; the fallback keeps the base
  features and is counted in ``pfer_base_fallback_phones``), else the zero vector exactly like the reference's unknown-phone
  branch (:130-137), counted in ``pfer_unknown_phones`` -- so a gap from the reference's panphon PFER is visible in the result;
* else NO features: PFER cannot be computed, ``evaluate_batch`` returns PER in the ``pfer`` slot, sets
  ``pfer_is_per_fallback`` and says so on stderr (once).

Segmentation: the table's ``ipa_segs`` when it has one and keeps every character (reference :41-47), otherwise the
reference's own Unicode rule (base character + combining marks and spacing modifier letters U+02B0-U+02FF, :49-64), which
satisfies all nine tokenisation assertions of the reference (:449-457).
"""
from __future__ import annotations

import csv
import os
import sys
import unicodedata
from typing import Dict, List, Optional, Sequence

import numpy as np

NUM_FEATURES = 24  # panphon's articulatory features


class CsvFeatureTable:
    """Minimal stand-in for panphon.FeatureTable over a panphon-format CSV (``ipa`` column + 24 feature columns)."""

    def __init__(self, path: str):
        self.vectors: Dict[str, List[int]] = {}
        with open(path, newline="", encoding="utf-8") as f:
            rows = csv.reader(f)
            header = next(rows)
            if len(header) != NUM_FEATURES + 1:
                raise ValueError(f"{path}: expected an 'ipa' column and {NUM_FEATURES} feature columns, got {len(header)}")
            sign = {"+": 1, "-": -1, "0": 0}
            for row in rows:
                if row:
                    self.vectors[unicodedata.normalize("NFD", row[0])] = [sign[v.strip()] for v in row[1:]]

        self.base_fallbacks: Dict[str, int] = {}  # segment -> times its base character's row stood in for it

    def word_to_vector_list(self, word: str, numeric: bool = True):
        key = unicodedata.normalize("NFD", word)
        v = self.vectors.get(key)
        if v is None and len(key) > 1:
            # a diacritic / modifier the table has no row for (ejective, nasalised vowel ...): the base character's features
            # instead of the all-zero vector.  panphon would additionally flip the features its diacritic_definitions name.
            base = "".join(ch for ch in key if not (unicodedata.category(ch).startswith("M") or unicodedata.category(ch) == "Lm"
                                                    or ch in "\u02d0\u02d1"))
            v = self.vectors.get(base) or (self.vectors.get(base[0]) if base else None)
            if v is not None:
                self.base_fallbacks[word] = self.base_fallbacks.get(word, 0) + 1
        return [list(v)] if v is not None else []


_ft = None
_ft_loaded = False
_warned = False


def _get_feature_table():
    """panphon.FeatureTable, else CsvFeatureTable(WIPA_PANPHON_CSV), else None (cached)."""
    global _ft, _ft_loaded
    if not _ft_loaded:
        _ft_loaded = True
        try:
            import panphon

            _ft = panphon.FeatureTable()
        except Exception:
            path = os.environ.get("WIPA_PANPHON_CSV")
            _ft = CsvFeatureTable(path) if path else None
    return _ft


def set_feature_table(ft) -> None:
    """Install a feature source (any object with ``word_to_vector_list(phone, numeric=True)`` and optionally
    ``ipa_segs(text)``); ``None`` re-runs the discovery."""
    global _ft, _ft_loaded, _pfer_calc, _pfer_calc_cosine
    _ft, _ft_loaded = ft, ft is not None
    _pfer_calc = _pfer_calc_cosine = None


def _unicode_segments(text: str) -> List[str]:
    segments: List[str] = []
    for ch in text:
        cat = unicodedata.category(ch)
        is_modifier = cat.startswith("M") or (cat == "Lm" and "ʰ" <= ch <= "˿")
        if segments and is_modifier:
            segments[-1] += ch
        else:
            segments.append(ch)
    return segments


def tokenize_ipa(text: str) -> List[str]:
    text = text.replace(" ", "")
    if not text:
        return []
    ft = _get_feature_table()
    if ft is not None and hasattr(ft, "ipa_segs"):
        phones = ft.ipa_segs(text)
        if "".join(phones) == text:  # the table's segmenter kept every character
            return list(phones)
    return _unicode_segments(text)


def normalize_ipa_for_comparison(text: str) -> str:
    text = unicodedata.normalize("NFC", text).replace(" ", "")
    return text.replace("g", "ɡ")  # Latin g -> IPA g


def edit_distance(a: Sequence, b: Sequence) -> int:
    """Levenshtein distance (substitution, insertion, deletion all cost 1) -- ``editdistance.eval`` of the reference."""
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


class PFERCalculator:
    """Phone Feature Error Rate, Hamming form (reference :108-213): edit distance with insertion / deletion cost 1 and
    substitution cost = (number of the 24 features that differ) / 24; identical phone strings cost 0; a phone the table
    does not know has the zero vector."""

    def __init__(self, ft=None):
        self.ft = ft if ft is not None else _get_feature_table()
        if self.ft is None:
            raise RuntimeError("PFER needs articulatory features: install panphon or point WIPA_PANPHON_CSV at its ipa_all.csv")
        self.num_features = NUM_FEATURES
        self.unknown_phones: Dict[str, int] = {}  # phone -> times it got the zero vector

    def get_phone_features(self, phone: str) -> np.ndarray:
        try:
            features = self.ft.word_to_vector_list(phone, numeric=True)
            if len(features) > 0:
                return np.array(features[0])
        except Exception:
            pass
        self.unknown_phones[phone] = self.unknown_phones.get(phone, 0) + 1
        return np.zeros(self.num_features)

    def feature_distance(self, phone1: str, phone2: str) -> float:
        if phone1 == phone2:
            return 0.0
        mismatches = np.sum(self.get_phone_features(phone1) != self.get_phone_features(phone2))
        return float(mismatches) / self.num_features

    def phone_feature_error_rate(self, reference: str, hypothesis: str) -> float:
        ref, hyp = tokenize_ipa(reference), tokenize_ipa(hypothesis)
        if not ref:
            return 0.0 if not hyp else 100.0
        m, n = len(ref), len(hyp)
        dp = np.zeros((m + 1, n + 1))
        dp[:, 0] = np.arange(m + 1)
        dp[0, :] = np.arange(n + 1)
        for i in range(1, m + 1):
            for j in range(1, n + 1):
                dp[i, j] = min(dp[i - 1, j] + 1.0, dp[i, j - 1] + 1.0, dp[i - 1, j - 1] + self.feature_distance(ref[i - 1], hyp[j - 1]))
        return dp[m, n] / m * 100.0


class PFERCalculatorCosine(PFERCalculator):
    """Cosine form (reference :216-287, Taguchi et al.'s LPhD_combined): equal feature vectors continue the diagonal for
    free; otherwise insertion, deletion and substitution all cost 1 - cos(ref features, hyp features)."""

    def get_phone_features(self, phone: str) -> np.ndarray:
        return super().get_phone_features(phone).astype(float)

    @staticmethod
    def cosine_distance(f1: np.ndarray, f2: np.ndarray) -> float:
        den = np.linalg.norm(f1) * np.linalg.norm(f2)
        if den == 0:
            den = 0.001
        return 1.0 - float(np.dot(f1, f2)) / den

    def phone_feature_error_rate(self, reference: str, hypothesis: str) -> float:
        ref, hyp = tokenize_ipa(reference), tokenize_ipa(hypothesis)
        if not ref:
            return 0.0 if not hyp else 100.0
        rf, hf = [self.get_phone_features(p) for p in ref], [self.get_phone_features(p) for p in hyp]
        m, n = len(ref), len(hyp)
        dp = np.zeros((m + 1, n + 1))
        dp[:, 0] = np.arange(m + 1)
        dp[0, :] = np.arange(n + 1)
        for i in range(1, m + 1):
            for j in range(1, n + 1):
                if np.array_equal(rf[i - 1], hf[j - 1]):
                    dp[i, j] = dp[i - 1, j - 1]
                else:
                    dp[i, j] = min(dp[i, j - 1], dp[i - 1, j], dp[i - 1, j - 1]) + self.cosine_distance(rf[i - 1], hf[j - 1])
        return dp[m, n] / m * 100.0


_pfer_calc: Optional[PFERCalculator] = None
_pfer_calc_cosine: Optional[PFERCalculatorCosine] = None


def get_pfer_calculator() -> PFERCalculator:
    global _pfer_calc
    if _pfer_calc is None:
        _pfer_calc = PFERCalculator()
    return _pfer_calc


def get_pfer_calculator_cosine() -> PFERCalculatorCosine:
    global _pfer_calc_cosine
    if _pfer_calc_cosine is None:
        _pfer_calc_cosine = PFERCalculatorCosine()
    return _pfer_calc_cosine


def phone_feature_error_rate(reference: str, hypothesis: str) -> float:
    return get_pfer_calculator().phone_feature_error_rate(reference, hypothesis)


def phone_feature_error_rate_cosine(reference: str, hypothesis: str) -> float:
    return get_pfer_calculator_cosine().phone_feature_error_rate(reference, hypothesis)


def evaluate_batch(references: List[str], hypotheses: List[str]) -> Dict:
    """reference :346-378: the RAW strings are scored (no normalisation here), per-sample scores are returned.
    Without a feature source the ``pfer`` slot carries PER and ``pfer_is_per_fallback`` is True (stderr says so once):
    best-checkpoint selection then ranks by PER."""
    global _warned
    assert len(references) == len(hypotheses), "Reference and hypothesis lists must have same length"
    have_features = _get_feature_table() is not None
    if not have_features and not _warned:
        _warned = True
        print("WARNING: no articulatory feature table (panphon / WIPA_PANPHON_CSV): PFER is NOT computed, the 'pfer' values below "
              "are PER and best-checkpoint selection ranks by PER", file=sys.stderr, flush=True)
    per_scores, pfer_scores = [], []
    calc = get_pfer_calculator() if have_features else None
    if calc is not None:
        calc.unknown_phones.clear()
        getattr(calc.ft, "base_fallbacks", {}).clear()
    for ref, hyp in zip(references, hypotheses):
        per = phone_error_rate(ref, hyp)
        per_scores.append(per)
        pfer_scores.append(phone_feature_error_rate(ref, hyp) if have_features else per)
    empty = not per_scores
    return {"per": 0.0 if empty else float(np.mean(per_scores)), "pfer": 0.0 if empty else float(np.mean(pfer_scores)),
            "per_std": 0.0 if empty else float(np.std(per_scores)), "pfer_std": 0.0 if empty else float(np.std(pfer_scores)),
            "num_samples": len(references), "per_scores": per_scores, "pfer_scores": pfer_scores,
            "pfer_is_per_fallback": not have_features,
            # phones the feature source did not know (scored with the zero vector, as the reference does) and, on the CSV path,
            # phones scored with their base character's features: both are where this PFER can differ from panphon's
            "pfer_unknown_phones": dict(calc.unknown_phones) if calc is not None else {},
            "pfer_base_fallback_phones": dict(getattr(calc.ft, "base_fallbacks", {})) if calc is not None else {}}
