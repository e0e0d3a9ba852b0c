"""Audio front-end with the call surface of ``mlx_whisper.audio`` used by the reference
(scripts/ipa_data_loader.py:14,48,80-82; scripts/transcribe_single.py:6,43-45):
``load_audio``, ``pad_or_trim``, ``log_mel_spectrogram`` and the constants.

``log_mel_spectrogram`` runs on the GPU (K1, csrc/logmel.hip).  ``load_audio`` reads WAV
files directly: the reference shells out to ffmpeg, which this image does not have.
"""
from __future__ import annotations

import wave
from typing import Union

import numpy as np
import torch

from . import _lib
from .runtime import device, dt_code, on_stream, ptr, sptr, stream_id

SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
CHUNK_LENGTH = 30
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE
N_FRAMES = N_SAMPLES // HOP_LENGTH
PADDED_FRAMES = N_FRAMES + 2  # one zero halo row either side (conv1 padding=1)

ArrayLike = Union[np.ndarray, torch.Tensor]


def load_audio(file: str, sr: int = SAMPLE_RATE) -> np.ndarray:
    """Decode a PCM WAV file to mono float32 in [-1, 1) at ``sr`` Hz (s16 -> /32768 like the
    reference's ffmpeg pipe).  Other containers need an external decoder."""
    with wave.open(file, "rb") as w:
        n_ch, width, rate, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if width == 2:
        a = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        a = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        a = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"unsupported WAV sample width {width}")
    if n_ch > 1:
        a = a.reshape(-1, n_ch).mean(axis=1)
    if rate != sr:
        a = _resample(a, rate, sr)
    return np.ascontiguousarray(a, dtype=np.float32)


def _resample(a: np.ndarray, src: int, dst: int, zero_crossings: int = 16) -> np.ndarray:
    """Windowed-sinc (Hann, ``zero_crossings`` lobes each side) resampling, evaluated per OUTPUT sample -- the polyphase
    form: output m sits at input position m*src/dst and sums the 2*zero_crossings/cutoff input samples around it, so the work
    is n_out x taps whatever the ratio (44.1 kHz -> 16 kHz is up 160 / down 441: the zero-stuffed form would convolve
    211 M samples with a 14 k-tap filter).  Host side, numpy, chunked to bound memory."""
    n_in = len(a)
    n_out = int(np.ceil(n_in * dst / src))
    if n_in == 0 or n_out == 0:
        return np.zeros(0, dtype=np.float32)
    cutoff = min(1.0, dst / src)              # of the input Nyquist: low-pass at the lower of the two rates
    half = zero_crossings / cutoff            # filter half-width in input samples
    k = np.arange(-int(np.ceil(half)), int(np.ceil(half)) + 1)
    x = np.asarray(a, dtype=np.float64)
    y = np.empty(n_out, dtype=np.float64)
    step = max(1, (1 << 22) // len(k))        # ~32 MB of float64 per chunk
    for m0 in range(0, n_out, step):
        m = np.arange(m0, min(m0 + step, n_out))
        pos = m * (src / dst)
        base = np.floor(pos).astype(np.int64)
        idx = base[:, None] + k[None, :]
        dt = pos[:, None] - idx                # distance of every tap from the output position, in input samples
        w = cutoff * np.sinc(cutoff * dt) * (0.5 + 0.5 * np.cos(np.pi * np.clip(dt / half, -1.0, 1.0)))
        ok = (idx >= 0) & (idx < n_in)
        y[m] = (w * np.where(ok, x[np.clip(idx, 0, n_in - 1)], 0.0)).sum(axis=1)
    return y.astype(np.float32)


def pad_or_trim(array: ArrayLike, length: int = N_SAMPLES, axis: int = -1) -> ArrayLike:
    """Zero-pad or cut ``axis`` to ``length`` (ipa_data_loader.py:80)."""
    if isinstance(array, torch.Tensor):
        n = array.shape[axis]
        if n > length:
            array = array.narrow(axis, 0, length)
        elif n < length:
            pad = [0, 0] * array.ndim
            pad[2 * (array.ndim - 1 - (axis % array.ndim)) + 1] = length - n
            array = torch.nn.functional.pad(array, pad)
        return array
    array = np.asarray(array)
    n = array.shape[axis]
    if n > length:
        array = array.take(indices=range(length), axis=axis)
    elif n < length:
        widths = [(0, 0)] * array.ndim
        widths[axis] = (0, length - n)
        array = np.pad(array, widths)
    return array


_tables = {}
_workspaces = {}


def _get_tables(n_mels: int) -> torch.Tensor:
    dev = device()
    key = (dev.index, n_mels)
    t = _tables.get(key)
    if t is None:
        L = _lib.lib()
        with on_stream() as s:
            t = torch.empty(L.wipa_logmel_tables_bytes(n_mels), dtype=torch.uint8, device=dev)
            _lib.check(L.wipa_logmel_init(ptr(t), n_mels, sptr(s)), "wipa_logmel_init")
        _tables[key] = t
    return t


def padded_mel_rows(batch: int) -> int:
    return batch * PADDED_FRAMES + 4


def log_mel_padded(audio: torch.Tensor, n_mels: int = 80, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """audio [B, 480000] f32 on the GPU -> padded mel [B*3002 + 4, n_mels] in ``dtype``
    (frame t of clip b at row b*3002 + t + 1): the layout the encoder consumes directly."""
    L = _lib.lib()
    assert audio.is_cuda and audio.dtype == torch.float32 and audio.dim() == 2 and audio.shape[1] == N_SAMPLES
    audio = audio.contiguous()
    B = audio.shape[0]
    tables = _get_tables(n_mels)
    with on_stream() as s:
        key = (audio.device.index, stream_id())
        ws = _workspaces.get(key)
        need = L.wipa_logmel_workspace_bytes(B, n_mels)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=audio.device)  # one per library stream
            _workspaces[key] = ws
        mel = torch.empty(padded_mel_rows(B), n_mels, dtype=dtype, device=audio.device)
        _lib.check(L.wipa_logmel(ptr(audio), B, n_mels, ptr(tables), ptr(mel), dt_code(dtype), ptr(ws), ws.numel(), sptr(s)),
                   "wipa_logmel")
    return mel


def log_mel_spectrogram(audio: ArrayLike, n_mels: int = 80, padding: int = 0) -> torch.Tensor:
    """mlx_whisper.audio.log_mel_spectrogram: [n] -> [n_frames, n_mels] f32 (time-major), or
    [B, n] -> [B, n_frames, n_mels].  Clips are processed in the 30 s window the reference
    always uses (``pad_or_trim`` first, ipa_data_loader.py:80-82); returned tensors live on the GPU."""
    if isinstance(audio, np.ndarray):
        audio = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32))
    if padding:
        audio = torch.nn.functional.pad(audio, (0, padding))
    single = audio.dim() == 1
    if single:
        audio = audio[None]
    if audio.shape[-1] != N_SAMPLES:
        raise ValueError(f"log_mel_spectrogram expects {N_SAMPLES} samples per clip (use pad_or_trim), got {audio.shape[-1]}")
    audio = audio.to(device=device(), dtype=torch.float32)
    B = audio.shape[0]
    padded = log_mel_padded(audio, n_mels, torch.float32)
    mel = padded[: B * PADDED_FRAMES].view(B, PADDED_FRAMES, n_mels)[:, 1 : N_FRAMES + 1]
    return mel[0] if single else mel
