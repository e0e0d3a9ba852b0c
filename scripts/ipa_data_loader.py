"""IPA dataset / batch builder with the surface of the reference's scripts/ipa_data_loader.py
(``IPADataset`` :17-131, ``create_data_loader`` :134-157), on top of whisper_ipa_amd.

Same JSON schema (``audio_path``, ``ipa_transcription``, ``speaker_id``, ``dataset_source``), same
batch dict keys (``mel_features [B,3000,n_mels]``, ``tokens [B,T]``, ``ipa_texts``, ``audio_paths``),
same token framing (SOT sequence incl. <|notimestamps|> + BPE(ipa) + EOT, padded with EOT).
Differences forced by the platform: audio is read as WAV/PCM (no ffmpeg), the log-mel of the
whole batch is ONE GPU launch, and the returned arrays are torch tensors on the GPU.
"""
from __future__ import annotations

import json
import os
import sys
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from whisper_ipa_amd.audio import N_SAMPLES, load_audio, log_mel_spectrogram, pad_or_trim  # noqa: E402


class IPADataset:
    """audio + IPA transcription pairs (reference :17-131)."""

    def __init__(self, json_path: str, tokenizer, n_mels: int = 80, audio_root: str = ""):
        self.json_path = Path(json_path)
        self.tokenizer = tokenizer
        self.n_mels = n_mels
        self.audio_root = audio_root
        with open(self.json_path) as f:
            self.data = json.load(f)
        print(f"Loaded {len(self.data)} samples from {self.json_path}")

    def __len__(self) -> int:
        return len(self.data)

    def __getitem__(self, idx: int) -> Dict:
        entry = self.data[idx]
        path = entry["audio_path"]
        audio = load_audio(os.path.join(self.audio_root, path) if self.audio_root else path)
        return {
            "audio": audio,
            "ipa_text": entry["ipa_transcription"],
            "audio_path": path,
            "metadata": {"speaker_id": entry.get("speaker_id", "unknown"),
                         "dataset_source": entry.get("dataset_source", "unknown")},
        }

    def tokenize_batch(self, ipa_texts: List[str]) -> torch.Tensor:
        """<|sot|><|en|><|transcribe|><|notimestamps|> ipa <|eot|>, EOT-padded to the longest row."""
        import ctypes as C

        from whisper_ipa_amd import _lib

        tok = self.tokenizer
        ids = [tok.encode(t) for t in ipa_texts]
        prefix = list(tok.sot_sequence_including_notimestamps)
        n = len(ids)
        flat = [i for row in ids for i in row]
        ld = len(prefix) + max(len(r) for r in ids) + 1
        out = torch.empty(n, ld, dtype=torch.int32)
        width = _lib.lib().wipa_build_token_batch((C.c_int32 * max(len(flat), 1))(*flat), (C.c_int32 * n)(*[len(r) for r in ids]), n,
                                                  (C.c_int32 * len(prefix))(*prefix), len(prefix), tok.eot, out.data_ptr(), ld)
        _lib.check(0 if width > 0 else width, "wipa_build_token_batch")
        return out[:, :width].contiguous()

    # reference name (ipa_data_loader.py:102)
    _tokenize_ipa_batch = tokenize_batch

    def get_batch(self, indices: List[int], audio_for: Optional[List[bool]] = None) -> Dict:
        """reference :63-100.  ``audio_for`` (one flag per index; None = all): read the audio and compute the log-mel only for
        the flagged clips -- ``mel_features`` then holds just those rows, in order (None when no clip is flagged).  The
        fine-tune loop passes the clips whose frozen-encoder features are not cached yet (training.FrozenFeatureCache), so a
        cached clip costs neither a file read nor a mel nor an encoder pass."""
        need = [True] * len(indices) if audio_for is None else [bool(f) for f in audio_for]
        assert len(need) == len(indices)
        entries = [self.data[i] for i in indices]
        samples = [self[i] for i, f in zip(indices, need) if f]
        mel = None
        if samples:
            audio = np.stack([pad_or_trim(s["audio"], N_SAMPLES) for s in samples])
            mel = log_mel_spectrogram(audio, n_mels=self.n_mels)  # [n_flagged, 3000, n_mels] on the GPU, one launch
        texts = [e["ipa_transcription"] for e in entries]
        tokens = self.tokenize_batch(texts)
        return {
            "mel_features": mel,
            "tokens": tokens.to(mel.device) if mel is not None else tokens,
            "ipa_texts": texts,
            "audio_paths": [e["audio_path"] for e in entries],
        }


def create_data_loader(json_path: str, multilingual: bool = True, n_mels: int = 80, audio_root: str = "",
                       allow_byte_fallback: bool = False) -> IPADataset:
    from whisper_ipa_amd import tokenizer as tok

    print(f"Loading Whisper tokenizer (multilingual={multilingual})...")
    tokenizer = tok.require_real_vocabulary(tok.get_tokenizer(multilingual=multilingual), allow_byte_fallback,
                                            "building IPA training / evaluation batches")
    tokenizer.language = "en"  # reference :152 (does not change the frozen sot_sequence)
    return IPADataset(json_path, tokenizer, n_mels=n_mels, audio_root=audio_root)
