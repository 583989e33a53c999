"""Portable counter-based synthetic weights (numpy statement).

No GLM-ASR checkpoint exists offline, so benchmarks and full-size parity
checks use weights from a generator that is restated bit-identically in three places:

  * here (numpy)                         -- used by oracle/gen_golden.py and the tests
  * oracle/sonic_oracle.c  synth_fill()  -- the CPU oracle
  * csrc/synth.hip        synth_fill_*   -- the engine, directly into HBM

Definition (all integer arithmetic mod 2^64, all float arithmetic IEEE fp32, one rounding
per operation, no fused multiply-add):

    h      = FNV-1a-64(name)                               (utf-8 bytes)
    key    = splitmix64_mix(seed * 0x9E3779B97F4A7C15 + h)
    z_i    = splitmix64_mix(key + (i + 1) * 0x9E3779B97F4A7C15)
    bits_i = z_i >> 40                                     (24 bits)
    r_i    = float(bits_i) * 2^-23 - 1.0                   (exact, in [-1, 1))
    v_i    = offset + r_i * scale                          (mul rounded, then add rounded)
    if bf16: v_i = round_to_nearest_even_bf16(v_i)

scale / offset per tensor kind are given by ``kind_params``.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np

from .spec import ModelDims, tensor_inventory

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _mix(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def tensor_key(seed: int, name: str) -> int:
    with np.errstate(over="ignore"):
        k = np.uint64(seed) * GOLDEN + np.uint64(fnv1a64(name))
        return int(_mix(np.asarray([k], dtype=np.uint64))[0])


def round_bf16(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 (round to nearest even) -> fp32.  Finite inputs only."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


def to_bf16_bits(x: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)
    return r.astype(np.uint16)


def round_f16(x: np.ndarray) -> np.ndarray:
    """fp32 -> IEEE half (round to nearest even) -> fp32."""
    return np.ascontiguousarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def synth_fill(seed: int, name: str, n: int, scale: float, offset: float, bf16) -> np.ndarray:
    key = np.uint64(tensor_key(seed, name))
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = _mix(key + idx * GOLDEN)
    bits = (z >> np.uint64(40)).astype(np.int32)
    r = bits.astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)
    v = r * np.float32(scale)
    v = np.float32(offset) + v
    v = v.astype(np.float32)
    if int(bf16) == 2:          # a bf16 checkpoint loaded with torch_dtype=float16 (the int8 mode, asr.py:156)
        return round_f16(round_bf16(v))
    return round_bf16(v) if bf16 else v


def kind_params(kind: str, shape: Tuple[int, ...]) -> Tuple[float, float]:
    """(scale, offset) as fp32-representable python floats."""
    if kind == "mat":
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        s = math.sqrt(3.0 / fan_in)
    elif kind == "embed":
        s = math.sqrt(3.0 / shape[1])
    elif kind == "bias":
        return (float(np.float32(0.1)), 0.0)
    elif kind == "norm":
        return (float(np.float32(0.1)), 1.0)
    else:
        raise ValueError(kind)
    return (float(np.float32(s)), 0.0)


def synth_state_dict(dims: ModelDims, seed: int, bf16) -> Dict[str, np.ndarray]:
    """All parameters as fp32 numpy arrays (bf16-representable when ``bf16``; fp16(bf16(.)) when ``bf16 == 2``)."""
    out: Dict[str, np.ndarray] = {}
    for name, shape, kind in tensor_inventory(dims):
        n = int(np.prod(shape))
        scale, offset = kind_params(kind, shape)
        out[name] = synth_fill(seed, name, n, scale, offset, bf16).reshape(shape)
    return out


def synth_pcm(i: int, n_samples: int) -> np.ndarray:
    """Synthetic 16 kHz int16 PCM for segment ``i``: 0.1*N(0,1) noise plus a
    220*(1 + i mod 7) Hz tone at 0.2 amplitude, clipped, rounded to int16."""
    rng = np.random.default_rng(1234 + i)
    x = 0.1 * rng.standard_normal(n_samples)
    t = np.arange(n_samples) / 16000.0
    x = x + 0.2 * np.sin(2 * np.pi * 220.0 * (1 + i % 7) * t)
    x = np.clip(x, -1.0, 32766.0 / 32767.0)
    return np.round(x * 32767.0).astype(np.int16)
