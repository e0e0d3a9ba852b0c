"""Inference entry point with the behaviour of the reference's scripts/transcribe_single.py:
base model -> fp32 -> overlay the fine-tuned ``decoder.*`` tensors of ``<checkpoint>/model.safetensors``
(reference :10-39), then load_audio -> pad_or_trim -> log_mel_spectrogram -> model.encoder -> greedy
``decode(language="en", without_timestamps=True)`` -> ``result[0].text.strip()`` (reference :41-56).

The reference hard-codes its three paths (:10,59-60) and fetches the base model by hub name; here
the same constants are the defaults and can be overridden on the command line, and the base
model is a local directory (config.json + weights.safetensors) because there is no network.
All compute runs in libwipa.so on the GPU.
"""
from __future__ import annotations

import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from whisper_ipa_amd.audio import load_audio, log_mel_spectrogram, pad_or_trim  # noqa: E402
from whisper_ipa_amd.decoding import DecodingOptions, decode  # noqa: E402
from whisper_ipa_amd.load_models import load_model, overlay_decoder_weights  # noqa: E402


def load_checkpoint_model(checkpoint_path: str, base_model: str = "mlx-community/whisper-large-v3-mlx"):
    print(f"Loading base model architecture: {base_model}")
    model = load_model(base_model)
    model.set_dtype(torch.float32)
    if checkpoint_path:
        try:
            n = overlay_decoder_weights(model, checkpoint_path)
        except FileNotFoundError as e:
            print(f"ERROR: {e}")
            sys.exit(1)
        print(f"Found {n} decoder parameters to load")
        print("✓ Decoder weights loaded successfully")
    return model


def transcribe_file(model, audio_path: str) -> str:
    print(f"Transcribing {audio_path}...")
    audio = pad_or_trim(load_audio(audio_path))
    mel = log_mel_spectrogram(audio, n_mels=model.dims.n_mels)[None].to(torch.float32)
    options = DecodingOptions(language="en", without_timestamps=True)  # IPA is decoded "as English"
    audio_features = model.encoder(mel)
    result = decode(model, audio_features, options)
    return result[0].text.strip()


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--checkpoint", default="checkpoints/whisper-ipa/checkpoint-8000")
    ap.add_argument("--audio", default="4.wav")
    ap.add_argument("--base-model", default="mlx-community/whisper-large-v3-mlx",
                    help="local directory with config.json + weights.safetensors (hub names cannot resolve offline)")
    ap.add_argument("--allow-byte-fallback", action="store_true",
                    help="run without the Whisper vocabulary (WIPA_TIKTOKEN unset): ids >= 256 print as <|idN|>; synthetic weights only")
    args = ap.parse_args(argv)
    from whisper_ipa_amd.tokenizer import get_tokenizer, require_real_vocabulary

    require_real_vocabulary(get_tokenizer(True), args.allow_byte_fallback, "transcribing with a trained checkpoint")
    model = load_checkpoint_model(args.checkpoint, args.base_model)
    text = transcribe_file(model, args.audio)
    print("\n" + "=" * 50)
    print(f"Audio: {args.audio}")
    print(f"Prediction: {text}")
    print("=" * 50)
    return text


if __name__ == "__main__":
    main()
