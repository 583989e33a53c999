"""sonicscribe_amd -- MI355X-native ASR engine behind SonicScribe's ``ASRModel.transcribe()``.

Only what the hot path needs lives here: host logic mirroring the reference interface
(``asr.ASRModel``, ``frontend``), the ctypes binding (``engine``) to the C-ABI library built
from ``csrc/`` (hand-written HIP for gfx950), and weight plumbing.
"""
__version__ = "0.1.0"
