"""sonicscribe_amd -- MI355X-native ASR engine behind SonicScribe's ``ASRModel.transcribe()``.

Only what the hot path needs lives here: host logic mirroring the reference interface
(``asr.ASRModel``, ``frontend``), the ctypes binding (``engine``) to the C-ABI library built
from ``csrc/`` (hand-written HIP for gfx950), and weight plumbing.
"""
__version__ = "0.1.0"

import os as _os

# More hardware queues for the HIP runtime than its default of 4 (read once, at the runtime's first call; a value the user set stays): an
# engine's handles own two streams each, and streams that share a hardware queue run in order (csrc/engine.cpp sonic_more_hw_queues has
# the measurement).  Set here as well as in the library's constructor so that a process which imports this package before it touches
# torch.cuda gets it even when torch initialises the runtime first.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
