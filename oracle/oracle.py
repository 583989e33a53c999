"""ctypes binding of the CPU oracle (oracle/sonic_oracle.c).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from
the product package sonicscribe_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsonic_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "sonic_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB_PATH


class Dims(C.Structure):
    _fields_ = [
        ("n_mels", C.c_int), ("n_frames", C.c_int), ("enc_T", C.c_int),
        ("enc_d", C.c_int), ("enc_ff", C.c_int), ("enc_layers", C.c_int), ("enc_heads", C.c_int), ("enc_rotary_dim", C.c_int),
        ("enc_theta", C.c_float), ("enc_ln_eps", C.c_float),
        ("merge", C.c_int),
        ("dec_d", C.c_int), ("dec_ff", C.c_int), ("dec_layers", C.c_int), ("dec_heads", C.c_int), ("dec_kv_heads", C.c_int), ("dec_head_dim", C.c_int),
        ("dec_theta", C.c_float), ("dec_rms_eps", C.c_float),
        ("vocab", C.c_int), ("audio_token_id", C.c_int), ("n_eos", C.c_int),
        ("eos", C.c_int * 8),
    ]


class Outputs(C.Structure):
    _fields_ = [
        ("conv1", C.c_void_p), ("conv2", C.c_void_p), ("enc_layers", C.c_void_p), ("enc_out", C.c_void_p),
        ("audio_embeds", C.c_void_p), ("dec_layers", C.c_void_p), ("prefill_logits", C.c_void_p),
        ("step_logits", C.c_void_p), ("new_ids", C.c_void_p), ("n_new", C.c_void_p), ("force_ids", C.c_void_p),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.oracle_set_threads.argtypes = [C.c_int]
        # cap the OpenMP team: on a 256-thread host the default team makes every small region cost milliseconds
        _lib.oracle_set_threads(int(os.environ.get("SONIC_ORACLE_THREADS", min(os.cpu_count() or 8, 48))))
        _lib.oracle_model_create.restype = C.c_void_p
        _lib.oracle_model_create.argtypes = [C.POINTER(Dims), C.POINTER(C.c_void_p), C.c_int, C.c_int]
        _lib.oracle_model_destroy.argtypes = [C.c_void_p]
        _lib.oracle_transcribe.restype = C.c_int
        _lib.oracle_transcribe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(Outputs)]
        _lib.oracle_transcribe_multi.restype = C.c_int
        _lib.oracle_transcribe_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(Outputs)]
        _lib.oracle_audio_features.restype = C.c_int
        _lib.oracle_audio_features.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(Outputs)]
        _lib.oracle_logmel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib.oracle_mel_filters.argtypes = [C.c_int, C.c_void_p]
        _lib.oracle_synth_fill.argtypes = [C.c_uint64, C.c_char_p, C.c_long, C.c_float, C.c_float, C.c_int, C.c_void_p]
        _lib.oracle_linear.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4
        _lib.oracle_gelu.argtypes = [C.c_void_p, C.c_long, C.c_int]
        _lib.oracle_attention.argtypes = [C.c_void_p] * 4 + [C.c_int] * 8
        _lib.oracle_rope.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int]
        _lib.oracle_layernorm.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float, C.c_int]
        _lib.oracle_rmsnorm.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_float, C.c_int]
        _lib.oracle_encoder_layer.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        _lib.oracle_quantize_rows.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_void_p]
        _lib.oracle_linear_int8.argtypes = [C.c_void_p] * 5 + [C.c_int] * 3
        _lib.oracle_round.restype = C.c_float
        _lib.oracle_round.argtypes = [C.c_float, C.c_int]
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_dims(d) -> Dims:
    """d: sonicscribe_amd.spec.ModelDims"""
    x = Dims()
    x.n_mels, x.n_frames, x.enc_T = d.n_mels, d.n_frames, d.enc_T
    x.enc_d, x.enc_ff, x.enc_layers, x.enc_heads, x.enc_rotary_dim = d.enc_d, d.enc_ff, d.enc_layers, d.enc_heads, d.enc_rotary_dim
    x.enc_theta, x.enc_ln_eps, x.merge = d.enc_rope_theta, d.enc_ln_eps, d.merge
    x.dec_d, x.dec_ff, x.dec_layers, x.dec_heads, x.dec_kv_heads, x.dec_head_dim = d.dec_d, d.dec_ff, d.dec_layers, d.dec_heads, d.dec_kv_heads, d.dec_head_dim
    x.dec_theta, x.dec_rms_eps = d.dec_rope_theta, d.dec_rms_eps
    x.vocab, x.audio_token_id, x.n_eos = d.vocab, d.audio_token_id, len(d.eos_ids)
    for i, e in enumerate(d.eos_ids):
        x.eos[i] = e
    return x


def logmel(pcm_i16: np.ndarray, n_mels: int = 128, n_frames: int = 3000):
    pcm = np.ascontiguousarray(pcm_i16, dtype=np.int16)
    feats = np.empty((n_mels, n_frames), np.float32)
    mask = np.empty(n_frames, np.int32)
    lib().oracle_logmel(_p(pcm), pcm.size, n_mels, n_frames, _p(feats), _p(mask))
    return feats, mask


def mel_filters(n_mels: int = 128) -> np.ndarray:
    out = np.empty((201, n_mels), np.float32)
    lib().oracle_mel_filters(n_mels, _p(out))
    return out


def synth_fill(seed: int, name: str, n: int, scale: float, offset: float, bf16) -> np.ndarray:
    """bf16: False/0 fp32, True/1 bf16-rounded, 2 = a bf16 checkpoint loaded as fp16 (the int8 mode's weights, asr.py:156)."""
    out = np.empty(n, np.float32)
    lib().oracle_synth_fill(seed, name.encode(), n, scale, offset, int(bf16), _p(out))
    return out


MODE_FP32, MODE_BF16, MODE_FP16, MODE_INT8 = 0, 1, 2, 3


def gelu(x: np.ndarray, bf16: bool = True) -> np.ndarray:
    """erf GELU in fp32 (the op order of torch GELU(approximate="none")), rounded to bf16 when asked."""
    y = np.array(x, np.float32, order="C")
    lib().oracle_gelu(_p(y), y.size, int(bf16))
    return y


def quantize_rows(w: np.ndarray):
    """Int8Params.cuda(): row-wise absmax int8 of a [N][K] fp16-valued matrix -> (CB int8 [N][K], SCB fp32 [N])."""
    w = np.ascontiguousarray(w, np.float32)
    cb = np.empty(w.shape, np.int8); scb = np.empty(w.shape[0], np.float32)
    lib().oracle_quantize_rows(_p(w), w.shape[0], w.shape[1], _p(cb), _p(scb))
    return cb, scb


def linear_int8(x: np.ndarray, cb: np.ndarray, scb: np.ndarray, bias=None) -> np.ndarray:
    """One Linear8bitLt(threshold=6.0) call on x [T][K] (fp16-valued fp32)."""
    x = np.ascontiguousarray(x, np.float32); cb = np.ascontiguousarray(cb, np.int8); scb = np.ascontiguousarray(scb, np.float32)
    b = np.ascontiguousarray(bias, np.float32) if bias is not None else None
    y = np.empty((x.shape[0], cb.shape[0]), np.float32)
    lib().oracle_linear_int8(_p(x), _p(cb), _p(scb), _p(b), _p(y), x.shape[0], cb.shape[0], x.shape[1])
    return y


def round_f16(x: np.ndarray) -> np.ndarray:
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


class Model:
    """Oracle model over a state dict {name: fp32 ndarray} in spec.tensor_inventory order."""

    def __init__(self, dims, state: Dict[str, np.ndarray], bf16: bool = False, mode: Optional[int] = None):
        """mode: MODE_FP32 / MODE_BF16 / MODE_FP16 / MODE_INT8 (fp16 activations + LLM.int8 linears); `bf16` is the old boolean."""
        from sonicscribe_amd.spec import tensor_inventory
        self.dims = dims
        self.mode = int(mode) if mode is not None else (MODE_BF16 if bf16 else MODE_FP32)
        self.bf16 = self.mode == MODE_BF16
        self._keep = [np.ascontiguousarray(state[name], dtype=np.float32) for name, _, _ in tensor_inventory(dims)]
        arr = (C.c_void_p * len(self._keep))(*[a.ctypes.data for a in self._keep])
        self._cd = make_dims(dims)
        self.h = lib().oracle_model_create(C.byref(self._cd), arr, len(self._keep), self.mode)
        if not self.h:
            raise RuntimeError("oracle_model_create failed")

    def close(self):
        if self.h:
            lib().oracle_model_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def transcribe(self, feats: np.ndarray, n_valid_frames, prompt, max_new: int, want=(), force_ids=None):
        """feats [n_mels][n_frames] with an int n_valid_frames, or (multi-window request) [W][n_mels][n_frames] with a list."""
        d = self.dims
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        nv = np.ascontiguousarray(np.atleast_1d(n_valid_frames), dtype=np.int32)
        W = int(nv.size)
        assert feats.size == W * d.n_mels * d.n_frames
        prompt = np.ascontiguousarray(prompt, dtype=np.int32)
        P = prompt.size
        n_audio_max = W * (d.enc_T // d.merge)
        bufs = {
            "conv1": np.zeros((d.enc_d, d.n_frames), np.float32) if "conv1" in want else None,
            "conv2": np.zeros((d.enc_d, d.enc_T), np.float32) if "conv2" in want else None,
            "enc_layers": np.zeros((d.enc_layers, d.enc_T, d.enc_d), np.float32) if "enc_layers" in want else None,
            "enc_out": np.zeros((d.enc_T, d.enc_d), np.float32) if "enc_out" in want else None,
            "audio_embeds": np.zeros((n_audio_max, d.dec_d), np.float32),
            "dec_layers": np.zeros((d.dec_layers, P, d.dec_d), np.float32) if "dec_layers" in want else None,
            "prefill_logits": np.zeros(d.vocab, np.float32),
            "step_logits": np.zeros((max_new, d.vocab), np.float32),
            "new_ids": np.zeros(max_new, np.int32),
        }
        n_new = C.c_int(0)
        o = Outputs()
        for k, v in bufs.items():
            setattr(o, k, v.ctypes.data if v is not None else None)
        o.n_new = C.addressof(n_new)
        if force_ids is not None:
            force_ids = np.ascontiguousarray(force_ids, dtype=np.int32)
            o.force_ids = force_ids.ctypes.data
        rc = lib().oracle_transcribe_multi(self.h, _p(feats), _p(nv), W, _p(prompt), P, int(max_new), C.byref(o))
        if rc != 0:
            raise ValueError("Audio features and audio tokens do not match")
        res = {k: v for k, v in bufs.items() if v is not None}
        res["n_new"] = n_new.value
        res["new_ids"] = bufs["new_ids"][: n_new.value]
        res["step_logits"] = bufs["step_logits"][: n_new.value]
        return res
