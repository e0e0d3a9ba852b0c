"""``load_model`` with the surface of ``mlx_whisper.load_models.load_model``
(scripts/train_whisper_ipa.py:499; scripts/transcribe_single.py:12; scripts/evaluate_model.py:34)
plus the checkpoint writer of ``save_checkpoint`` (scripts/train_whisper_ipa.py:410-443).

The reference resolves hub NAMES (``mlx-community/whisper-*-mlx``); there is no network here, so a
model is a LOCAL directory with ``config.json`` (the ModelDimensions fields) and
``weights.safetensors`` / ``model.safetensors`` in mlx_whisper's flat dotted key names
(Linear ``[out,in]``, Conv1d ``[C_out, K, C_in]``).  OpenAI / HF layouts are converted by key rename.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

import torch

from .whisper import ModelDimensions, Whisper

_DIM_KEYS = ["n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer", "n_vocab", "n_text_ctx",
             "n_text_state", "n_text_head", "n_text_layer"]

NAMED_DIMS = {
    "tiny": ModelDimensions(80, 1500, 384, 6, 4, 51865, 448, 384, 6, 4),
    "base": ModelDimensions(80, 1500, 512, 8, 6, 51865, 448, 512, 8, 6),
    "small": ModelDimensions(80, 1500, 768, 12, 12, 51865, 448, 768, 12, 12),
    "medium": ModelDimensions(80, 1500, 1024, 16, 24, 51865, 448, 1024, 16, 24),
    "large-v3": ModelDimensions(128, 1500, 1280, 20, 32, 51866, 448, 1280, 20, 32),
}


def dims_from_name(name: str) -> Optional[ModelDimensions]:
    """'mlx-community/whisper-small-mlx' -> small dims (n_mels rule: train_whisper_ipa.py:517)."""
    low = name.lower()
    for key in ("large-v3", "medium", "small", "base", "tiny"):
        if key in low:
            return NAMED_DIMS[key]
    return None


def _hf_to_mlx_key(k: str) -> Optional[str]:
    """transformers WhisperForConditionalGeneration key -> mlx_whisper key."""
    if not k.startswith("model."):
        return None
    k = k[len("model."):]
    rep = [("encoder.layers.", "encoder.blocks."), ("decoder.layers.", "decoder.blocks."),
           (".self_attn_layer_norm.", ".attn_ln."), (".encoder_attn_layer_norm.", ".cross_attn_ln."),
           (".final_layer_norm.", ".mlp_ln."), (".self_attn.", ".attn."), (".encoder_attn.", ".cross_attn."),
           (".q_proj.", ".query."), (".k_proj.", ".key."), (".v_proj.", ".value."), (".out_proj.", ".out."),
           (".fc1.", ".mlp1."), (".fc2.", ".mlp2."), ("encoder.layer_norm.", "encoder.ln_post."),
           ("decoder.layer_norm.", "decoder.ln."), ("decoder.embed_tokens.", "decoder.token_embedding."),
           ("decoder.embed_positions.weight", "decoder.positional_embedding")]
    for a, b in rep:
        k = k.replace(a, b)
    if k.startswith("encoder.embed_positions") or k.endswith("key.bias"):
        return None
    return k


def convert_weights(flat: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Accept mlx_whisper keys as they are; rename HF keys and permute HF conv weights
    ``[C_out, C_in, K]`` -> ``[C_out, K, C_in]``."""
    if any(k.startswith("model.encoder.") for k in flat):
        out = {}
        for k, v in flat.items():
            nk = _hf_to_mlx_key(k)
            if nk is None:
                continue
            if nk in ("encoder.conv1.weight", "encoder.conv2.weight"):
                v = v.permute(0, 2, 1).contiguous()
            out[nk] = v
        return out
    return dict(flat)


def load_safetensors(path: str) -> Dict[str, torch.Tensor]:
    from safetensors.torch import load_file

    return load_file(path, device="cpu")


def save_safetensors(path: str, flat: Dict[str, torch.Tensor]) -> None:
    """mlx ``save_safetensors`` counterpart (train_whisper_ipa.py:422)."""
    from safetensors.torch import save_file

    save_file({k: v.detach().to("cpu").contiguous() for k, v in flat.items()}, path)


def load_model(path_or_name: str, dtype: torch.dtype = torch.float32, sinusoid_rounding: str = "f32") -> Whisper:
    """Local directory -> Whisper.  Raises with a clear message for hub names (offline).
    ``sinusoid_rounding="fp16"`` reproduces mlx_whisper's fp16-built encoder sinusoid table (Whisper.__init__)."""
    if not os.path.isdir(path_or_name):
        raise FileNotFoundError(
            f"load_model: '{path_or_name}' is not a local directory. The reference downloads hub names "
            "(mlx-community/whisper-*-mlx); this build is offline: pass a directory holding config.json and "
            "weights.safetensors / model.safetensors in mlx_whisper key names.")
    cfg_path = os.path.join(path_or_name, "config.json")
    if os.path.exists(cfg_path):
        cfg = json.load(open(cfg_path))
        dims = ModelDimensions(**{k: int(cfg[k]) for k in _DIM_KEYS})
    else:
        dims = dims_from_name(os.path.basename(os.path.normpath(path_or_name)))
        if dims is None:
            raise FileNotFoundError(f"load_model: no config.json in {path_or_name} and the name gives no size")
    weights = None
    for fn in ("weights.safetensors", "model.safetensors"):
        p = os.path.join(path_or_name, fn)
        if os.path.exists(p):
            weights = convert_weights(load_safetensors(p))
            break
    if weights is None:
        raise FileNotFoundError(f"load_model: no weights.safetensors / model.safetensors in {path_or_name}")
    model = Whisper(dims, dtype=dtype, sinusoid_rounding=sinusoid_rounding)
    model.load_weights(weights)
    return model


def save_model(model: Whisper, path: str) -> None:
    """Write config.json + weights.safetensors so load_model(path) round-trips."""
    os.makedirs(path, exist_ok=True)
    json.dump({k: getattr(model.dims, k) for k in _DIM_KEYS}, open(os.path.join(path, "config.json"), "w"), indent=2)
    save_safetensors(os.path.join(path, "weights.safetensors"), model.flat_parameters())


def overlay_decoder_weights(model: Whisper, checkpoint_dir: str) -> int:
    """transcribe_single.py:15-33: keep the base model, overwrite the ``decoder.*`` tensors found
    in ``<checkpoint_dir>/model.safetensors``.  Returns how many tensors were replaced."""
    p = os.path.join(checkpoint_dir, "model.safetensors")
    if not os.path.exists(p):
        raise FileNotFoundError(f"No weights found at {p}")
    trained = load_safetensors(p)
    dec = {k: v for k, v in trained.items() if k.startswith("decoder.")}
    known = set(model.flat_parameters())
    dec = {k: v for k, v in dec.items() if k in known}
    model.load_weights(dec, strict=False)
    return len(dec)
