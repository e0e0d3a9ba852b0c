"""whisper_ipa_amd -- MI355X-native Whisper speech-to-IPA hot path.

Python host side mirroring the ``mlx_whisper`` surface the reference scripts call
(SURVEY.md section 8b); all arithmetic is in hand-written HIP kernels behind the C ABI of
``libwipa.so`` (include/wipa.h).  Importing this package never touches the GPU; any compute
entry point raises if the extension is not built or no GPU is present -- there is no CPU
fallback.
"""
from . import _lib  # noqa: F401
from . import runtime as _runtime

_runtime.request_hw_queues()  # GPU_MAX_HW_QUEUES=8 unless the user set it; must precede the first HIP call (see runtime.py)

from . import audio, decoding, pipeline, tokenizer, whisper  # noqa: F401,E402
from .audio import load_audio, log_mel_spectrogram, pad_or_trim  # noqa: F401
from .decoding import DecodingOptions, DecodingResult, decode  # noqa: F401
from .pipeline import PassResult, TranscribePipeline, transcribe_batches  # noqa: F401
from .whisper import ModelDimensions, Whisper  # noqa: F401

__version__ = "0.1.0"
