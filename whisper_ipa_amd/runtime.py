"""Plumbing shared by the Python host side: the library stream, dtype codes, pointers.

torch is used for device memory, streams and (later) autograd glue only; every
computation on the path goes through libwipa.so.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import Optional

import torch

from . import _lib

_streams = {}


def device() -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.WipaError("whisper_ipa_amd needs a GPU (MI355X); there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def stream(dev: Optional[torch.device] = None) -> torch.cuda.Stream:
    """One dedicated (non-default, capturable) HIP stream per device for all libwipa work."""
    dev = dev or device()
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _streams.get(key)
    if s is None:
        s = torch.cuda.Stream(device=key)
        _streams[key] = s
    return s


@contextlib.contextmanager
def on_stream():
    """Run torch allocations/copies and libwipa launches on the library stream, ordered
    after the caller's stream on entry and before it on exit."""
    s = stream()
    cur = torch.cuda.current_stream()
    s.wait_stream(cur)
    with torch.cuda.stream(s):
        yield s
    cur.wait_stream(s)


def dt_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return _lib.WIPA_F32
    if dtype == torch.bfloat16:
        return _lib.WIPA_BF16
    raise _lib.WipaError(f"unsupported dtype {dtype}: libwipa computes in float32 or bfloat16")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr()


def sptr(s: torch.cuda.Stream) -> int:
    return s.cuda_stream


def ptr_table(tensors) -> "C.Array":
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr
