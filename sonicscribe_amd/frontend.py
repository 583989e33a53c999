"""Host-side restatement of the reference's pre-/post-steps around the device path.

Everything here is cheap host bookkeeping; the arithmetic-heavy
rows (a5-a12) run in the HIP engine.  numpy only -- no torch, no GPU.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from .spec import ModelDims, FULL, audio_token_count, valid_frames

BASE_INSTRUCTION = "Please transcribe this audio into text"  # asr.py:375


def pcm_bytes_to_float(audio_data: bytes) -> np.ndarray:
    """transcription_manager.py:45-54: bytes -> int16 -> float32 / 32768 -> [1, N]."""
    a = np.frombuffer(audio_data, dtype=np.int16)
    return (a.astype(np.float32) / np.float32(32768.0))[None, :]


def normalise_to_int16(wav: np.ndarray) -> np.ndarray:
    """asr.py:247-276 without the disk: first channel, peak-normalise when max|x| > 1e-6, then
    the PCM_16 temp-WAV round trip.

    ``sf.write(path, float32, sr)`` picks subtype PCM_16 for ``.wav`` and libsndfile converts
    float -> short as ``lrintf(x * 0x7FFF)`` (normalised float mode, no clipping needed since
    |x| <= 1 after the peak normalisation; third-party behaviour of soundfile==0.13.1 /
    libsndfile, which is absent offline -- SURVEY.md §8a row a2 marks it "verify when
    soundfile is available").  HF ``load_audio`` then reads the shorts back as ``s / 32768``,
    which the device kernel applies when it loads the int16 PCM.
    """
    w = np.asarray(wav, dtype=np.float32)
    if w.ndim == 2:
        w = w[0]
    w = np.ascontiguousarray(w)
    if w.size == 0:
        return np.zeros(0, np.int16)
    m = np.float32(np.max(np.abs(w)))
    if m > np.float32(1e-6):
        w = (w / m).astype(np.float32)
    p = (w * np.float32(32767.0)).astype(np.float32)
    q = np.rint(p)  # round-half-even, as lrintf in the default rounding mode
    return np.clip(q, -32768, 32767).astype(np.int16)


def format_hotwords_prompt(hotwords: Optional[Sequence[str]], max_hotwords: int = 10) -> str:
    """asr.py:303-333.  The reference de-duplicates the RAW strings through ``set()`` (asr.py:318-322) and only then strips and
    lower-cases them, so ``["Alpha", "alpha "]`` yields two entries, ``"alpha", "alpha"``; only exact repeats of a raw string
    collapse.  ``set()`` iterates in an order that changes from process to process (string hash randomisation); this restatement
    keeps the first-seen order of the raw strings, which is one of the orders the reference can produce."""
    if not hotwords:
        return ""
    raw = []
    for hw in hotwords:
        if hw not in raw:            # set(hotwords): exact duplicates of the raw value only
            raw.append(hw)
    cleaned = [hw.strip().lower() for hw in raw if hw and isinstance(hw, str) and hw.strip()]
    if not cleaned:
        return ""
    cleaned = cleaned[:max_hotwords]
    return ". Pay special attention to these important terms: " + ", ".join(f'"{h}"' for h in cleaned)


def build_instruction(hotwords: Optional[Sequence[str]]) -> str:
    return BASE_INSTRUCTION + format_hotwords_prompt(hotwords or [])


def split_windows(n_samples: int, dims: ModelDims = FULL, max_audio_len: int = 655) -> List[Tuple[int, int]]:
    """HF:processing_glmasr.py:136-157: cut one audio into <=30 s windows (at most 21)."""
    win = dims.n_samples
    max_windows = int(max_audio_len // dims.chunk_seconds)
    n_win = max(1, (n_samples + win - 1) // win)
    n_win = min(n_win, max_windows)
    cap = min(n_samples, n_win * win)
    return [(i * win, min((i + 1) * win, cap)) for i in range(n_win)]


def request_audio_tokens(n_samples: int, dims: ModelDims = FULL) -> Tuple[int, List[int]]:
    """(placeholder count in the prompt, per-window kept rows).

    The processor counts placeholders from the *summed* valid frames of all windows
    (processing_glmasr.py:166-169) while the model keeps ``post_len`` rows per window
    (modeling_glmasr.py:399-406); they agree for every single-window request.
    """
    wins = split_windows(n_samples, dims)
    frames = [valid_frames(e - s, dims) for s, e in wins]
    total = audio_token_count(sum(frames), dims.merge)
    per_win = [max(0, audio_token_count(f, dims.merge)) for f in frames]
    return total, per_win


def max_new_tokens_committed(segment_duration: float) -> int:
    """transcription_manager.py:37."""
    return min(50 + int(segment_duration * 5), 200)


def resample_sinc_hann(wav: np.ndarray, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> np.ndarray:
    """asr.py:255-261: ``torchaudio.transforms.Resample(orig, new)`` with its defaults (sinc_interp_hann, width 6,
    rolloff 0.99), restated in numpy.  Never taken on the reference's own call sites (both pass 16 kHz); torchaudio is
    absent offline, so this branch is unpinned."""
    import math
    wav = np.asarray(wav, dtype=np.float32)
    if orig_freq == new_freq or wav.size == 0:
        return wav
    g = math.gcd(int(orig_freq), int(new_freq))
    of, nf = int(orig_freq) // g, int(new_freq) // g
    base = min(of, nf) * rolloff
    width = math.ceil(lowpass_filter_width * of / base)
    idx = np.arange(-width, width + of, dtype=np.float64)[None, :] / of
    t = (np.arange(0, -nf, -1, dtype=np.float64)[:, None] / nf + idx) * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    with np.errstate(divide="ignore", invalid="ignore"):
        kern = np.where(t == 0, 1.0, np.sin(t) / t)
    kern = (kern * window * (base / of)).astype(np.float32)          # [nf][2*width + of]
    length = wav.shape[-1]
    x = np.pad(wav, (width, width + of))
    n_out = (x.size - kern.shape[1]) // of + 1
    frames = np.lib.stride_tricks.as_strided(x, shape=(n_out, kern.shape[1]), strides=(x.strides[0] * of, x.strides[0]))
    out = (frames @ kern.T).reshape(-1)
    target = int(math.ceil(nf * length / of))
    return out[:target].astype(np.float32)
