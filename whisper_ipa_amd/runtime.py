"""Plumbing shared by the Python host side: library streams, dtype codes, pointers.

torch is used for device memory, streams and (later) autograd glue only; every
computation on the path goes through libwipa.so.

Streams: all libwipa work runs on dedicated non-default HIP streams (graph capture needs
one).  Stream 0 is the default library stream; ``use_stream(i)`` selects another one for a
block of code so independent clip sub-batches can run concurrently (the compute-bound
encoder of one sub-batch overlaps the HBM/latency-bound decode loop of another).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import threading
from typing import Optional

import torch

from . import _lib

_streams = {}
_pad_streams = []  # WIPA_PAD_STREAMS: idle streams created before the library's own (kept alive)
_own_streams = {}  # device -> the library's own streams 0 .. OWN_STREAM_COUNT - 1, created together
OWN_STREAM_COUNT = 8
_stream_cus = {}  # library stream id -> CU limit (limit_stream_cus), read when the stream is first used
_tls = threading.local()


ROCM_DEFAULT_HW_QUEUES = 4
_hwq = {"requested_in_time": None}


def request_hw_queues(n: int = 8) -> int:
    """ROCm multiplexes the HIP streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the variable
    ONCE, when the HIP runtime starts.  Several passes in flight (pipeline.transcribe_batches) want 8: with 4 passes on 4 queues
    a pass takes 86-88 ms, on 8 queues 72 ms (whisper-small, 64 clips).  Called when the package is imported:
    if the user has not set the variable and nothing has initialised the GPU yet, set it; a user's own value is never
    overridden.  Returns the count that will be (or is) in effect.
    LIMIT of what the host can know: "nothing has initialised the GPU yet" is judged by torch.cuda.is_initialized(), but a bare
    torch.cuda.is_available() / device_count() can start the HIP runtime without setting that flag -- import the package before
    ANY torch.cuda call (bench.py, the scripts and __graft_entry__.smoke() do) and hw_queues() is right."""
    if "GPU_MAX_HW_QUEUES" in os.environ:
        if _hwq["requested_in_time"] is None:
            _hwq["requested_in_time"] = not torch.cuda.is_initialized()  # set by the user (or an earlier import) before HIP started?
        return hw_queues()
    if torch.cuda.is_initialized():
        _hwq["requested_in_time"] = False  # too late: the runtime has read its flags
        return ROCM_DEFAULT_HW_QUEUES
    os.environ["GPU_MAX_HW_QUEUES"] = str(int(n))
    _hwq["requested_in_time"] = True
    return int(n)


def hw_queues() -> int:
    """the hardware-queue count in effect for this process, as far as the host can know it: GPU_MAX_HW_QUEUES when it was in
    the environment before the HIP runtime started, the ROCm default otherwise"""
    v = os.environ.get("GPU_MAX_HW_QUEUES")
    if v is None or _hwq["requested_in_time"] is False:
        return ROCM_DEFAULT_HW_QUEUES
    try:
        return max(1, int(v))
    except ValueError:
        return ROCM_DEFAULT_HW_QUEUES


def device() -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.WipaError("whisper_ipa_amd needs a GPU (MI355X); there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def stream_id() -> int:
    return getattr(_tls, "sid", 0)


def stream(sid: Optional[int] = None) -> torch.cuda.Stream:
    """The library stream ``sid`` (default: the one selected by use_stream, else 0) of the
    current device."""
    sid = stream_id() if sid is None else sid
    key = (torch.cuda.current_device(), sid)
    s = _streams.get(key)
    if s is None:
        n_cus = _stream_cus.get(sid)
        if n_cus is None and os.environ.get("WIPA_OWN_STREAMS", "1") == "1":
            # The library's OWN HIP streams (wipa_stream_create), not torch's pool -- and the first OWN_STREAM_COUNT of them created
            # TOGETHER at the first request, before any of them carries work, each used once so that its hardware queue exists.
            # ROCm hands hardware queues to streams in creation order: created together, the streams of up to four passes in
            # flight sit on distinct queues even with ROCm's default of four (72.0 / 71.3 ms per pass against 72.1 / 71.8 / 71.3
            # with eight queues), whereas torch's pool streams on four queues share them (86-88 ms), and a stream created WHILE
            # others carry work can land on a busy queue (the passes then run one after another: 85 ms on the decode loops).
            # DESIGN.md 8.2, profiles/r05_queue_pad_sweep.txt.  WIPA_OWN_STREAMS=0 restores torch's pool (A/B);
            # WIPA_PAD_STREAMS=n creates n idle streams first (experiments).
            def make():
                raw = C.c_void_p()
                _lib.check(_lib.lib().wipa_stream_create(C.byref(raw)), "wipa_stream_create")
                st = torch.cuda.ExternalStream(raw.value, device=key[0])  # lives as long as the process
                with torch.cuda.stream(st):
                    torch.zeros(1, device=torch.device("cuda", key[0]))
                return st
            if not _own_streams.get(key[0]):
                for _ in range(int(os.environ.get("WIPA_PAD_STREAMS", "0"))):
                    _pad_streams.append(make())
                _own_streams[key[0]] = [make() for _ in range(int(os.environ.get("WIPA_OWN_STREAM_COUNT", str(OWN_STREAM_COUNT))))]
                torch.cuda.synchronize(key[0])
            own = _own_streams[key[0]]
            s = own[sid] if 0 <= sid < len(own) else make()  # ids beyond the eager set: on demand (no pass in flight uses one)
        elif n_cus is None:
            s = torch.cuda.Stream(device=key[0])
        else:
            raw = C.c_void_p()
            _lib.check(_lib.lib().wipa_stream_create_cu_limited(int(n_cus), C.byref(raw)), "wipa_stream_create_cu_limited")
            s = torch.cuda.ExternalStream(raw.value, device=key[0])  # lives as long as the process
        _streams[key] = s
    return s


def limit_stream_cus(sid: int, n_cus: Optional[int]) -> None:
    """Library stream ``sid`` (not yet used) will run its kernels on the first ``n_cus`` CUs only (wipa_stream_create_cu_limited)."""
    if any(k[1] == sid for k in _streams):
        raise _lib.WipaError(f"library stream {sid} already exists; set its CU limit before first use")
    if n_cus is None:
        _stream_cus.pop(sid, None)
    else:
        _stream_cus[sid] = int(n_cus)


@contextlib.contextmanager
def use_stream(sid: int):
    """Run the enclosed libwipa calls on library stream ``sid`` WITHOUT ordering them against
    the caller's stream or other library streams (the caller synchronises, e.g. through
    decoding.greedy_collect)."""
    prev = (getattr(_tls, "sid", 0), getattr(_tls, "detached", False))
    _tls.sid, _tls.detached = sid, True
    try:
        with torch.cuda.stream(stream(sid)):
            yield stream(sid)
    finally:
        _tls.sid, _tls.detached = prev


@contextlib.contextmanager
def on_stream():
    """Run torch allocations/copies and libwipa launches on the current library stream.
    Outside use_stream() the work is ordered after the caller's stream on entry and the
    caller's stream waits for it on exit."""
    s = stream()
    if getattr(_tls, "detached", False):
        with torch.cuda.stream(s):
            yield s
        return
    cur = torch.cuda.current_stream()
    if cur != s:
        s.wait_stream(cur)
    with torch.cuda.stream(s):
        yield s
    if cur != s:
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
