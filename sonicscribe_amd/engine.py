"""ctypes binding of libsonic_hip.so (include/sonic_hip.h).

The product path fails loudly when the HIP library is missing or no GPU is visible; there is no
CPU fallback (the CPU oracle lives under oracle/ and is test infrastructure only).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .spec import ModelDims

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libsonic_hip.so")

EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_SWIGLU = 0, 1, 2, 3
MODE_NATIVE, MODE_INT8, MODE_F16, MODE_F32 = 0, 1, 2, 3
DTYPE_F32, DTYPE_BF16 = 0, 1
SONIC_ERR_MISMATCH, SONIC_ERR_UNSUPPORTED = 4, 5

EXPORTS = [
    "sonic_device_count", "sonic_create", "sonic_destroy", "sonic_last_error", "sonic_load_tensor", "sonic_load_synthetic",
    "sonic_finalize_weights", "sonic_weight_bytes", "sonic_logmel", "sonic_encode", "sonic_transcribe_batch", "sonic_stage_pcm",
    "sonic_run_staged", "sonic_fetch_tokens", "sonic_get_timings", "sonic_synchronize", "sonic_test_gemm", "sonic_test_skinny",
    "sonic_test_attention", "sonic_test_decode_attention", "sonic_test_layernorm", "sonic_bench_gemm", "sonic_bench_skinny", "sonic_set_option", "sonic_debug_read", "sonic_debug_ktrace", "sonic_test_skinny_gu",
    "sonic_set_forced_ids", "sonic_test_greedy", "sonic_test_linear_int8",
    "sonic_ring_create", "sonic_ring_destroy", "sonic_ring_append", "sonic_ring_head", "sonic_transcribe_mixed", "sonic_stage_mixed",
    "sonic_prefill", "sonic_decode_step", "sonic_device_info", "sonic_memory_info",
    "sonic_abi_version", "sonic_slot_create", "sonic_slot_count", "sonic_run_staged_async", "sonic_wait",
    "sonic_service_begin", "sonic_service_end", "sonic_splice_rows", "sonic_service_step", "sonic_fetch_row", "sonic_fetch_rows", "sonic_prefill_enqueue",
    "sonic_runtime_info", "sonic_engine_info",
    "sonic_dispatch_create", "sonic_dispatch_submit", "sonic_dispatch_cancel", "sonic_dispatch_next", "sonic_dispatch_stats", "sonic_dispatch_close", "sonic_dispatch_destroy",
    "sonic_pipeline_create", "sonic_pipeline_submit", "sonic_pipeline_wait", "sonic_pipeline_stats", "sonic_pipeline_last_error", "sonic_pipeline_destroy",
]
ABI_VERSION = 7


class SonicDims(C.Structure):
    _fields_ = [
        ("n_mels", C.c_int32), ("n_frames", C.c_int32), ("enc_T", C.c_int32),
        ("enc_d", C.c_int32), ("enc_ff", C.c_int32), ("enc_layers", C.c_int32), ("enc_heads", C.c_int32), ("enc_rotary_dim", C.c_int32),
        ("enc_theta", C.c_float), ("enc_ln_eps", C.c_float),
        ("merge", C.c_int32),
        ("dec_d", C.c_int32), ("dec_ff", C.c_int32), ("dec_layers", C.c_int32), ("dec_heads", C.c_int32), ("dec_kv_heads", C.c_int32), ("dec_head_dim", C.c_int32),
        ("dec_theta", C.c_float), ("dec_rms_eps", C.c_float),
        ("vocab", C.c_int32), ("audio_token_id", C.c_int32), ("n_eos", C.c_int32),
        ("eos", C.c_int32 * 8),
    ]


class SonicTimings(C.Structure):
    _fields_ = [
        ("mel_ms", C.c_float), ("encoder_ms", C.c_float), ("prefill_ms", C.c_float), ("decode_ms", C.c_float), ("total_ms", C.c_float),
        ("gemm_ms", C.c_float), ("gemm_launches", C.c_int32), ("gemm_flops", C.c_double), ("decode_steps", C.c_int32),
        ("enc_gemm_ms", C.c_float), ("enc_gemm_flops", C.c_double),
        ("host_prefill_enqueue_ms", C.c_float), ("host_decode_launch_ms", C.c_float), ("host_decode_wait_ms", C.c_float), ("host_decode_launches", C.c_int32),
        ("decode_lookahead", C.c_int32), ("decode_launches_per_layer", C.c_int32),
    ]


def make_dims(d: ModelDims) -> SonicDims:
    x = SonicDims()
    x.n_mels, x.n_frames, x.enc_T = d.n_mels, d.n_frames, d.enc_T
    x.enc_d, x.enc_ff, x.enc_layers, x.enc_heads, x.enc_rotary_dim = d.enc_d, d.enc_ff, d.enc_layers, d.enc_heads, d.enc_rotary_dim
    x.enc_theta, x.enc_ln_eps, x.merge = d.enc_rope_theta, d.enc_ln_eps, d.merge
    x.dec_d, x.dec_ff, x.dec_layers, x.dec_heads, x.dec_kv_heads, x.dec_head_dim = d.dec_d, d.dec_ff, d.dec_layers, d.dec_heads, d.dec_kv_heads, d.dec_head_dim
    x.dec_theta, x.dec_rms_eps = d.dec_rope_theta, d.dec_rms_eps
    x.vocab, x.audio_token_id, x.n_eos = d.vocab, d.audio_token_id, len(d.eos_ids)
    for i, e in enumerate(d.eos_ids):
        x.eos[i] = e
    return x


_lib = None


def load_library():
    """Load libsonic_hip.so; raise RuntimeError (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"sonicscribe_amd: HIP extension not built ({LIB_PATH} missing). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C sonicscribe_amd/csrc`. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, ip, i64p, fp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.c_void_p
    lib.sonic_device_count.restype = C.c_int
    lib.sonic_create.restype = C.c_int
    lib.sonic_create.argtypes = [C.POINTER(SonicDims), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.sonic_destroy.argtypes = [vp]
    lib.sonic_destroy.restype = None
    lib.sonic_last_error.restype = C.c_char_p
    lib.sonic_last_error.argtypes = [vp]
    lib.sonic_load_tensor.argtypes = [vp, C.c_char_p, vp, C.c_int, i64p, C.c_int]
    lib.sonic_load_synthetic.argtypes = [vp, C.c_uint64]
    lib.sonic_finalize_weights.argtypes = [vp]
    lib.sonic_weight_bytes.restype = C.c_int64
    lib.sonic_weight_bytes.argtypes = [vp]
    lib.sonic_logmel.argtypes = [vp, vp, vp, C.c_int, vp, vp]
    lib.sonic_encode.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, vp]
    lib.sonic_transcribe_batch.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp]
    lib.sonic_stage_pcm.argtypes = [vp, vp, vp, C.c_int]
    lib.sonic_ring_create.argtypes = [vp, C.c_int64, C.POINTER(vp)]
    lib.sonic_ring_destroy.argtypes = [vp]
    lib.sonic_ring_destroy.restype = None
    lib.sonic_ring_append.argtypes = [vp, vp, C.c_int64, C.POINTER(C.c_int64)]
    lib.sonic_ring_head.argtypes = [vp]
    lib.sonic_ring_head.restype = C.c_int64
    lib.sonic_transcribe_mixed.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp]
    lib.sonic_stage_mixed.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, vp, C.c_int]
    lib.sonic_run_staged.argtypes = [vp, vp, C.c_int, vp, vp, vp, C.c_int]
    lib.sonic_fetch_tokens.argtypes = [vp, vp, C.c_int, vp, vp]
    lib.sonic_get_timings.argtypes = [vp, C.POINTER(SonicTimings)]
    lib.sonic_synchronize.argtypes = [vp]
    lib.sonic_test_gemm.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.sonic_test_skinny.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int]
    lib.sonic_test_attention.argtypes = [vp, vp, vp, vp, vp] + [C.c_int] * 7
    lib.sonic_test_decode_attention.argtypes = [vp, vp, vp, vp, vp] + [C.c_int] * 4
    lib.sonic_test_layernorm.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, C.c_int]
    lib.sonic_bench_gemm.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.sonic_bench_skinny.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.sonic_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    lib.sonic_debug_ktrace.argtypes = [vp, C.c_void_p, C.c_int64]
    lib.sonic_debug_read.argtypes = [vp, C.c_char_p, C.c_int, vp, C.c_int64]
    lib.sonic_test_skinny_gu.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int]
    lib.sonic_prefill.argtypes = [vp, vp, C.c_int, vp, vp, vp, C.c_int]
    lib.sonic_decode_step.argtypes = [vp, C.c_int, ip, ip]
    lib.sonic_prefill_enqueue.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    lib.sonic_device_info.argtypes = [C.c_int, C.c_char_p, C.c_int, i64p, i64p, ip]
    lib.sonic_runtime_info.argtypes = [C.c_int, ip, ip, ip]
    lib.sonic_pipeline_create.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.sonic_pipeline_submit.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, i64p]
    lib.sonic_pipeline_wait.argtypes = [vp, C.c_int64]
    lib.sonic_pipeline_stats.argtypes = [vp, i64p, i64p, ip]
    lib.sonic_pipeline_last_error.argtypes = [vp]
    lib.sonic_pipeline_last_error.restype = C.c_char_p
    lib.sonic_pipeline_destroy.argtypes = [vp]
    lib.sonic_memory_info.argtypes = [vp, i64p, i64p]
    lib.sonic_set_forced_ids.argtypes = [vp, vp, C.c_int, C.c_int]
    lib.sonic_test_greedy.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.sonic_test_linear_int8.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.sonic_slot_create.argtypes = [vp, C.POINTER(vp)]
    lib.sonic_slot_count.argtypes = [vp]
    lib.sonic_dispatch_create.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.POINTER(vp)]
    lib.sonic_dispatch_submit.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, i64p]
    lib.sonic_dispatch_cancel.argtypes = [vp, C.c_int64]
    lib.sonic_dispatch_next.argtypes = [vp, C.c_int, i64p, ip, vp, C.c_int, ip, C.c_char_p, C.c_int]
    lib.sonic_dispatch_stats.argtypes = [vp, i64p, i64p, ip, ip]
    lib.sonic_dispatch_close.argtypes = [vp]
    lib.sonic_dispatch_destroy.argtypes = [vp]
    lib.sonic_engine_info.argtypes = [vp, ip, ip, ip, ip, C.POINTER(vp)]
    lib.sonic_run_staged_async.argtypes = [vp, vp, C.c_int, vp, vp, vp, C.c_int]
    lib.sonic_wait.argtypes = [vp, C.c_int, ip]
    lib.sonic_service_begin.argtypes = [vp]
    lib.sonic_service_end.argtypes = [vp]
    lib.sonic_splice_rows.argtypes = [vp, vp, C.c_int, vp, vp, i64p]
    lib.sonic_service_step.argtypes = [vp, C.c_int, C.c_int, vp, vp, i64p, ip]
    lib.sonic_fetch_row.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.sonic_fetch_rows.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int]
    for name in EXPORTS:
        getattr(lib, name)
    if lib.sonic_abi_version() != ABI_VERSION:
        raise RuntimeError(f"sonicscribe_amd: {LIB_PATH} has ABI version {lib.sonic_abi_version()}, this binding expects {ABI_VERSION}: rebuild the library")
    _lib = lib
    return lib


class SonicError(RuntimeError):
    pass


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class RingSlice:
    """Samples [start, start + n) of a device ring: a decode window that never visits the host."""
    __slots__ = ("ring", "start", "n")

    def __init__(self, ring: "Ring", start: int, n: int):
        self.ring, self.start, self.n = ring, int(start), int(n)

    def __len__(self):
        return self.n


class Ring:
    """Raw wire PCM (int16) of one streaming session in HBM (sonic_ring_*)."""

    def __init__(self, engine: "Engine", capacity_samples: int):
        self.engine, self.capacity = engine, int(capacity_samples)
        h = C.c_void_p()
        engine._check(engine.lib.sonic_ring_create(engine.h, self.capacity, C.byref(h)))
        self.h = h
        if not hasattr(engine, "_rings"):
            engine._rings = []
        engine._rings.append(self)

    def append(self, pcm) -> int:
        """pcm: bytes (little-endian int16, as on the wire) or an int16 array.  Returns the absolute index of its first sample."""
        a = np.frombuffer(pcm, dtype=np.int16) if isinstance(pcm, (bytes, bytearray, memoryview)) else np.ascontiguousarray(pcm, dtype=np.int16)
        first = C.c_int64(0)
        rc = self.engine.lib.sonic_ring_append(self.h, _p(a) if a.size else None, a.size, C.byref(first))
        if rc != 0:
            raise RuntimeError(f"sonic_ring_append failed with status {rc}: " + (self.engine.lib.sonic_last_error(None) or b"").decode())
        return int(first.value)

    @property
    def head(self) -> int:
        return int(self.engine.lib.sonic_ring_head(self.h))

    def slice(self, start: int, n: int) -> RingSlice:
        return RingSlice(self, start, n)

    def close(self):
        if getattr(self, "h", None):
            self.engine.lib.sonic_ring_destroy(self.h)
            self.h = None
            if self in getattr(self.engine, "_rings", []):
                self.engine._rings.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One model replica on one MI355X."""

    def __init__(self, dims: ModelDims, device_id: int = 0, mode: int = MODE_NATIVE, max_batch: int = 32, max_ctx: int = 1024, _slot_of: Optional["Engine"] = None):
        self.lib = load_library()
        self.dims = dims
        self.max_batch, self.max_ctx = max_batch, max_ctx
        self._cd = make_dims(dims)
        self.owner: Optional["Engine"] = _slot_of       # a slot keeps its weight owner alive
        self._slots: List["Engine"] = []
        h = C.c_void_p()
        if _slot_of is not None:
            rc = self.lib.sonic_slot_create(_slot_of.h, C.byref(h))
        else:
            rc = self.lib.sonic_create(C.byref(self._cd), device_id, mode, max_batch, max_ctx, C.byref(h))
        if rc != 0:
            msg = (self.lib.sonic_last_error(None) or b"").decode()
            if rc == SONIC_ERR_UNSUPPORTED:
                raise ImportError(msg)
            raise (ValueError if "mode must be" in msg else SonicError)(msg)
        self.h = h

    @property
    def root(self) -> "Engine":
        return self.owner if self.owner is not None else self

    def slot(self) -> "Engine":
        """Another batch in flight on this engine's weights (sonic_slot_create): an Engine of its own in every respect - stream, buffers,
        KV cache, graphs, lock - that shares the owner's weight allocations.  Closed with its owner at the latest."""
        root = self.root
        s = Engine(self.dims, 0, 0, self.max_batch, self.max_ctx, _slot_of=root)
        root._slots.append(s)
        return s

    def slot_count(self) -> int:
        return int(self.lib.sonic_slot_count(self.h))

    def info(self) -> dict:
        """sonic_engine_info: what the library says about this handle (row / context capacity, mode, device, identity of its weight copy)."""
        v = [C.c_int32() for _ in range(4)]
        w = C.c_void_p()
        self._check(self.lib.sonic_engine_info(self.h, *[C.byref(x) for x in v], C.byref(w)))
        return {"max_batch": v[0].value, "max_ctx": v[1].value, "mode": v[2].value, "device": v[3].value, "weights_id": w.value}

    # -- plumbing
    def _check(self, rc: int):
        if rc != 0:
            msg = (self.lib.sonic_last_error(self.h) or b"").decode()
            if rc == SONIC_ERR_MISMATCH:
                raise ValueError(msg)
            raise SonicError(msg)

    def close(self):
        if getattr(self, "h", None):
            for s in list(self._slots):                      # slots read this engine's weights: they go first
                s.close()
            for r in list(getattr(self, "_rings", ())):      # rings belong to their engine and go first
                r.close()
            self.lib.sonic_destroy(self.h)
            self.h = None
            if self.owner is not None and self in self.owner._slots:
                self.owner._slots.remove(self)

    def ring_create(self, capacity_samples: int) -> "Ring":
        """Device-resident PCM ring of one streaming session (include/sonic_hip.h sonic_ring_*)."""
        return Ring(self, capacity_samples)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights
    def load_tensor(self, name: str, arr: np.ndarray, bf16_bits: bool = False):
        arr = np.ascontiguousarray(arr)
        shape = (C.c_int64 * arr.ndim)(*arr.shape)
        if bf16_bits:
            assert arr.dtype == np.uint16
            dt = DTYPE_BF16
        else:
            arr = arr.astype(np.float32, copy=False)
            dt = DTYPE_F32
        self._check(self.lib.sonic_load_tensor(self.h, name.encode(), _p(arr), dt, shape, arr.ndim))

    def load_state_dict(self, state: Dict[str, np.ndarray]):
        for k, v in state.items():
            self.load_tensor(k, v)
        self.finalize()

    def load_synthetic(self, seed: int):
        self._check(self.lib.sonic_load_synthetic(self.h, seed))
        self.finalize()

    def finalize(self):
        self._check(self.lib.sonic_finalize_weights(self.h))

    def weight_bytes(self) -> int:
        return int(self.lib.sonic_weight_bytes(self.h))

    # -- stages
    @staticmethod
    def _pack_pcm(segments: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
        offs = np.zeros(len(segments) + 1, np.int64)
        for i, s in enumerate(segments):
            offs[i + 1] = offs[i] + len(s)
        pcm = np.concatenate([np.ascontiguousarray(s, dtype=np.int16) for s in segments]) if segments else np.zeros(0, np.int16)
        if pcm.size == 0:
            pcm = np.zeros(1, np.int16)
        return np.ascontiguousarray(pcm), offs

    def logmel(self, segments: Sequence[np.ndarray]):
        d = self.dims
        pcm, offs = self._pack_pcm(segments)
        B = len(segments)
        feats = np.empty((B, d.n_mels, d.n_frames), np.float32)
        mask = np.empty((B, d.n_frames), np.int32)
        self._check(self.lib.sonic_logmel(self.h, _p(pcm), _p(offs), B, _p(feats), _p(mask)))
        return feats, mask

    def encode(self, feats: np.ndarray, n_valid_frames: Sequence[int], want_layers: bool = False, want_enc_out: bool = False):
        d = self.dims
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        B = feats.shape[0]
        nv = np.ascontiguousarray(n_valid_frames, dtype=np.int32)
        emb = np.empty((B, d.max_audio_tokens, d.dec_d), np.float32)
        n_audio = np.empty(B, np.int32)
        layers = np.empty((B, d.enc_layers, d.enc_T, d.enc_d), np.float32) if want_layers else None
        enc_out = np.empty((B, d.enc_T, d.enc_d), np.float32) if want_enc_out else None
        self._check(self.lib.sonic_encode(self.h, _p(feats), _p(nv), B, _p(emb), _p(n_audio), _p(layers), _p(enc_out)))
        return emb, n_audio, layers, enc_out

    # -- the hot call
    @staticmethod
    def _pack_prompts(prompts: Sequence[Sequence[int]]):
        offs = np.zeros(len(prompts) + 1, np.int64)
        for i, p in enumerate(prompts):
            offs[i + 1] = offs[i] + len(p)
        ids = np.concatenate([np.asarray(p, np.int32) for p in prompts]).astype(np.int32)
        return np.ascontiguousarray(ids), offs

    def _pack_mixed(self, segments):
        """Windows that are RingSlice objects stay on the device; the rest is packed like _pack_pcm (ring windows: empty host ranges)."""
        W = len(segments)
        host = [np.zeros(0, np.int16) if isinstance(s, RingSlice) else s for s in segments]
        pcm, offs = self._pack_pcm(host)
        rings = (C.c_void_p * W)(*[s.ring.h if isinstance(s, RingSlice) else None for s in segments])
        start = np.array([s.start if isinstance(s, RingSlice) else 0 for s in segments], np.int64)
        n = np.array([s.n if isinstance(s, RingSlice) else 0 for s in segments], np.int32)
        for s in segments:
            if isinstance(s, RingSlice) and s.ring.engine.root is not self.root:
                raise ValueError("a ring slice can only be decoded by the engine that owns the ring (or by a slot of it)")
        return pcm, offs, rings, start, n

    def transcribe_batch(self, segments: Sequence[Any], prompts: Sequence[Sequence[int]], max_new: Sequence[int],
                         req_win: Optional[Sequence[int]] = None, want_logits: bool = False):
        """segments: int16 PCM windows (<= 30 s each, already peak-normalised) or RingSlice objects (raw wire PCM resident in a device
        ring; normalised on the device over the windows of their request); one prompt per request. Returns (ids list, logits or None)."""
        if any(isinstance(s, RingSlice) for s in segments):
            pcm, offs, rings, start, n = self._pack_mixed(segments)
            ids, poffs = self._pack_prompts(prompts)
            R = len(prompts)
            mn = np.ascontiguousarray(max_new, dtype=np.int32)
            out_ld = int(mn.max())
            out = np.zeros((R, out_ld), np.int32)
            out_len = np.zeros(R, np.int32)
            rw = np.ascontiguousarray(req_win, dtype=np.int32) if req_win is not None else None
            logits = np.zeros((out_ld, R, self.dims.vocab), np.float32) if want_logits else None
            self._check(self.lib.sonic_transcribe_mixed(self.h, _p(pcm), _p(offs), rings, _p(start), _p(n), len(segments), _p(rw), R, _p(ids), _p(poffs),
                                                        _p(mn), _p(out), out_ld, _p(out_len), _p(logits)))
            return [out[r, : out_len[r]].copy() for r in range(R)], logits
        pcm, offs = self._pack_pcm(segments)
        ids, poffs = self._pack_prompts(prompts)
        R = len(prompts)
        mn = np.ascontiguousarray(max_new, dtype=np.int32)
        out_ld = int(mn.max())
        out = np.zeros((R, out_ld), np.int32)
        out_len = np.zeros(R, np.int32)
        rw = np.ascontiguousarray(req_win, dtype=np.int32) if req_win is not None else None
        logits = np.zeros((out_ld, R, self.dims.vocab), np.float32) if want_logits else None
        self._check(self.lib.sonic_transcribe_batch(self.h, _p(pcm), _p(offs), len(segments), _p(rw), R, _p(ids), _p(poffs), _p(mn),
                                                    _p(out), out_ld, _p(out_len), _p(logits)))
        return [out[r, : out_len[r]].copy() for r in range(R)], logits

    def stage_pcm(self, segments: Sequence[Any], req_win: Optional[Sequence[int]] = None):
        if any(isinstance(s, RingSlice) for s in segments):
            pcm, offs, rings, start, n = self._pack_mixed(segments)
            rw = np.ascontiguousarray(req_win, dtype=np.int32) if req_win is not None else None
            R = len(rw) - 1 if rw is not None else len(segments)
            self._check(self.lib.sonic_stage_mixed(self.h, _p(pcm), _p(offs), rings, _p(start), _p(n), len(segments), _p(rw), R))
            return
        pcm, offs = self._pack_pcm(segments)
        self._check(self.lib.sonic_stage_pcm(self.h, _p(pcm), _p(offs), len(segments)))

    def run_staged(self, prompts: Sequence[Sequence[int]], max_new: Sequence[int], req_win: Optional[Sequence[int]] = None,
                   want_logits: bool = False):
        ids, poffs = self._pack_prompts(prompts)
        mn = np.ascontiguousarray(max_new, dtype=np.int32)
        rw = np.ascontiguousarray(req_win, dtype=np.int32) if req_win is not None else None
        self._run_cache = (ids, poffs, mn, rw)
        self._check(self.lib.sonic_run_staged(self.h, _p(rw), len(prompts), _p(ids), _p(poffs), _p(mn), int(want_logits)))

    def prefill(self, prompts: Sequence[Sequence[int]], max_new: Sequence[int], req_win: Optional[Sequence[int]] = None, want_logits: bool = False, wait: bool = True):
        """Stage entry point: everything up to and including the first greedy token of the staged batch (sonic_prefill).  wait=False: the
        work is only queued when the call returns (sonic_prefill_enqueue; a following splice_rows orders itself behind it on the device)."""
        ids, poffs = self._pack_prompts(prompts)
        mn = np.ascontiguousarray(max_new, dtype=np.int32)
        rw = np.ascontiguousarray(req_win, dtype=np.int32) if req_win is not None else None
        if not wait:
            self._check(self.lib.sonic_prefill_enqueue(self.h, _p(rw), len(prompts), _p(ids), _p(poffs), _p(mn)))
            return
        self._check(self.lib.sonic_prefill(self.h, _p(rw), len(prompts), _p(ids), _p(poffs), _p(mn), int(want_logits)))

    def decode_step(self, n_steps: int = 1):
        """Stage entry point: up to n_steps further greedy steps; returns (rows still active, steps actually run)."""
        na, done = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.sonic_decode_step(self.h, int(n_steps), C.byref(na), C.byref(done)))
        return int(na.value), int(done.value)

    def memory_info(self):
        """(allocated, reserved) bytes of this handle's live device allocations (equal: there is no caching layer under the engine)."""
        a, r = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.sonic_memory_info(self.h, C.byref(a), C.byref(r)))
        return int(a.value), int(r.value)

    def rerun_staged(self):
        """Repeat the last run_staged call without re-packing (benchmark inner loop)."""
        ids, poffs, mn, rw = self._run_cache
        self._check(self.lib.sonic_run_staged(self.h, _p(rw), len(mn), _p(ids), _p(poffs), _p(mn), 0))

    def run_staged_async(self, prompts: Optional[Sequence[Sequence[int]]] = None, max_new: Optional[Sequence[int]] = None,
                         req_win: Optional[Sequence[int]] = None):
        """sonic_run_staged_async: returns at once, a worker thread of the handle runs the batch; wait() collects its status.
        Without arguments: the arguments of the last run_staged call."""
        if prompts is not None:
            ids, poffs = self._pack_prompts(prompts)
            mn = np.ascontiguousarray(max_new, dtype=np.int32)
            rw = np.ascontiguousarray(req_win, dtype=np.int32) if req_win is not None else None
            self._run_cache = (ids, poffs, mn, rw)
        ids, poffs, mn, rw = self._run_cache
        rc = self.lib.sonic_run_staged_async(self.h, _p(rw), len(mn), _p(ids), _p(poffs), _p(mn), 0)
        if rc != 0:
            raise SonicError((self.lib.sonic_last_error(None) or b"").decode() or f"sonic_run_staged_async failed with status {rc}")

    # -- continuous decoding (sonic_service_*): this engine's rows are a pool, requests prefilled on a slot are spliced in row by row
    def service_begin(self):
        self._check(self.lib.sonic_service_begin(self.h))

    def service_end(self):
        self._check(self.lib.sonic_service_end(self.h))

    def splice_rows(self, src: "Engine", src_rows: Sequence[int], dst_rows: Sequence[int]) -> int:
        """rows src_rows of `src` (requests of its last prefill()) -> free rows dst_rows of this engine; returns the chunk sequence number
        after which service_step()'s flags describe the new occupants"""
        a = np.ascontiguousarray(src_rows, dtype=np.int32); b = np.ascontiguousarray(dst_rows, dtype=np.int32)
        seq = C.c_int64(0)
        self._check(self.lib.sonic_splice_rows(self.h, src.h, len(a), _p(a), _p(b), C.byref(seq)))
        return int(seq.value)

    def service_step(self, n_chunks: int = 1, rows: int = 0):
        """queue n_chunks more chunks over rows 0 .. rows-1 (rounded up to 16; 0 = all); returns (finished[64], n_new[64], seq, n_active) of
        the newest completed check"""
        fin = np.zeros(64, np.int32); nn = np.zeros(64, np.int32)
        seq, na = C.c_int64(0), C.c_int32(0)
        self._check(self.lib.sonic_service_step(self.h, int(n_chunks), int(rows), _p(fin), _p(nn), C.byref(seq), C.byref(na)))
        return fin, nn, int(seq.value), int(na.value)

    def fetch_row(self, row: int, n: int) -> np.ndarray:
        out = np.zeros(max(1, int(n)), np.int32)
        self._check(self.lib.sonic_fetch_row(self.h, int(row), int(n), _p(out)))
        return out[:n].copy()

    def fetch_rows(self, rows: Sequence[int], counts: Sequence[int]) -> List[np.ndarray]:
        """fetch_row for several finished rows in one call (one wait, one release launch)"""
        r, c = np.asarray(rows, np.int32), np.asarray(counts, np.int32)
        ld = max(1, int(c.max()) if len(c) else 1)
        out = np.zeros((len(r), ld), np.int32)
        self._check(self.lib.sonic_fetch_rows(self.h, len(r), _p(r), _p(c), _p(out), ld))
        return [out[i, :int(c[i])].copy() for i in range(len(r))]

    def synchronize(self):
        """block until everything queued on this handle's stream has completed (sonic_synchronize)"""
        self._check(self.lib.sonic_synchronize(self.h))

    def wait(self, block: bool = True) -> bool:
        """Collect the asynchronous run (raises its error).  block=False: returns False while it is still running."""
        busy = C.c_int32(0)
        self._check(self.lib.sonic_wait(self.h, int(block), C.byref(busy)))
        return not busy.value

    def fetch_tokens(self, R: int, out_ld: int):
        out = np.zeros((R, out_ld), np.int32)
        out_len = np.zeros(R, np.int32)
        self._check(self.lib.sonic_fetch_tokens(self.h, _p(out), out_ld, _p(out_len), None))
        return [out[r, : out_len[r]].copy() for r in range(R)]

    def timings(self) -> Dict[str, float]:
        t = SonicTimings()
        self._check(self.lib.sonic_get_timings(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in SonicTimings._fields_}

    # -- kernel test hooks
    def test_gemm(self, A, W, bias=None, resid=None, epi=EPI_BIAS):
        A = np.ascontiguousarray(A, np.float32); W = np.ascontiguousarray(W, np.float32)
        M, K = A.shape; N = W.shape[0]
        n_out = N // 2 if epi == EPI_SWIGLU else N
        out = np.empty((M, n_out), np.float32)
        b = np.ascontiguousarray(bias, np.float32) if bias is not None else None
        r = np.ascontiguousarray(resid, np.float32) if resid is not None else None
        self._check(self.lib.sonic_test_gemm(self.h, _p(A), _p(W), _p(b), _p(r), _p(out), M, N, K, epi))
        return out

    def test_skinny(self, X, W):
        X = np.ascontiguousarray(X, np.float32); W = np.ascontiguousarray(W, np.float32)
        M, K = X.shape; N = W.shape[0]
        out = np.empty((M, N), np.float32)
        self._check(self.lib.sonic_test_skinny(self.h, _p(X), _p(W), _p(out), M, N, K))
        return out

    def test_attention(self, q, k, v, causal: bool):
        """q [B][Tq][Hq][hd], k/v [B][Tk][Hkv][hd] -> [B][Tq][Hq][hd]"""
        q = np.ascontiguousarray(q, np.float32); k = np.ascontiguousarray(k, np.float32); v = np.ascontiguousarray(v, np.float32)
        B, Tq, Hq, hd = q.shape; Tk, Hkv = k.shape[1], k.shape[2]
        out = np.empty_like(q)
        self._check(self.lib.sonic_test_attention(self.h, _p(q), _p(k), _p(v), _p(out), B, Tq, Tk, Hq, Hkv, hd, int(causal)))
        return out

    def test_decode_attention(self, q, k, v):
        """q [B][Hq][128], k/v [B][Tk][Hkv][128] -> [B][Hq][128]"""
        q = np.ascontiguousarray(q, np.float32); k = np.ascontiguousarray(k, np.float32); v = np.ascontiguousarray(v, np.float32)
        B, Hq, _ = q.shape; Tk, Hkv = k.shape[1], k.shape[2]
        out = np.empty_like(q)
        self._check(self.lib.sonic_test_decode_attention(self.h, _p(q), _p(k), _p(v), _p(out), B, Tk, Hq, Hkv))
        return out

    def test_layernorm(self, x, w, b=None, eps=1e-5, rms=False):
        x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32)
        bb = np.ascontiguousarray(b, np.float32) if b is not None else None
        out = np.empty_like(x)
        self._check(self.lib.sonic_test_layernorm(self.h, _p(x), _p(w), _p(bb), _p(out), x.shape[0], x.shape[1], eps, int(rms)))
        return out

    def bench_gemm(self, M: int, N: int, K: int, epi: int = EPI_BIAS_GELU, iters: int = 20) -> float:
        ms = C.c_float(0)
        self._check(self.lib.sonic_bench_gemm(self.h, M, N, K, epi, iters, C.byref(ms)))
        return float(ms.value)


def _bench_skinny(self, M: int, N: int, K: int, variant: int, iters: int = 50) -> float:
    us = C.c_float(0)
    self._check(self.lib.sonic_bench_skinny(self.h, M, N, K, variant, iters, C.byref(us)))
    return float(us.value)


def _set_option(self, key: str, value: int):
    self._check(self.lib.sonic_set_option(self.h, key.encode(), value))


def _debug_ktrace(self) -> np.ndarray:
    """[slot 8][block 512][point 8] device wall-clock ticks (10 ns) of the decode kernels of the layer set by option "ktrace"."""
    out = np.zeros((8, 512, 8), np.int64)
    self._check(self.lib.sonic_debug_ktrace(self.h, _p(out), out.size))
    return out


def _debug_read(self, name: str, shape, index: int = 0) -> np.ndarray:
    out = np.empty(shape, np.float32)
    self._check(self.lib.sonic_debug_read(self.h, name.encode(), index, _p(out), out.size))
    return out


def _test_skinny_gu(self, X, Wi):
    X = np.ascontiguousarray(X, np.float32); Wi = np.ascontiguousarray(Wi, np.float32)
    M, K = X.shape; N = Wi.shape[0]
    out = np.empty((M, N // 2), np.float32)
    self._check(self.lib.sonic_test_skinny_gu(self.h, _p(X), _p(Wi), _p(out), M, N, K))
    return out


def _set_forced_ids(self, ids):
    """ids: [R][ld] int array (token n of request r) or None to clear; see sonic_set_forced_ids."""
    if ids is None:
        self._check(self.lib.sonic_set_forced_ids(self.h, None, 0, 0))
        return
    a = np.ascontiguousarray(ids, dtype=np.int32)
    assert a.ndim == 2
    self._check(self.lib.sonic_set_forced_ids(self.h, _p(a), a.shape[0], a.shape[1]))


def _test_greedy(self, slabs, B: int, want_logits: bool = False):
    """slabs: [ksplit][mpad][V] fp32 -> (token per row [B], bf16 logits [B][V] or None)"""
    s = np.ascontiguousarray(slabs, np.float32)
    ks, mpad, V = s.shape
    tok = np.zeros(B, np.int32)
    lg = np.zeros((B, V), np.float32) if want_logits else None
    self._check(self.lib.sonic_test_greedy(self.h, _p(s), ks, mpad, V, B, _p(tok), _p(lg)))
    return tok, lg


def _test_linear_int8(self, X, W, bias=None, resid=None, group_rows=None, epi=EPI_BIAS):
    """One Linear8bitLt call; X [M][K], W [N][K] fp16-valued. group_rows: rows per reference call (default: all rows one call)."""
    X = np.ascontiguousarray(X, np.float32); W = np.ascontiguousarray(W, np.float32)
    M, K = X.shape; N = W.shape[0]
    out = np.empty((M, N // 2 if epi == EPI_SWIGLU else N), np.float32)
    b = np.ascontiguousarray(bias, np.float32) if bias is not None else None
    r = np.ascontiguousarray(resid, np.float32) if resid is not None else None
    self._check(self.lib.sonic_test_linear_int8(self.h, _p(X), _p(W), _p(b), _p(r), _p(out), M, N, K, int(group_rows or M), epi))
    return out


Engine.test_linear_int8 = _test_linear_int8
Engine.set_forced_ids = _set_forced_ids
Engine.test_greedy = _test_greedy
Engine.test_skinny_gu = _test_skinny_gu
Engine.debug_read = _debug_read
Engine.debug_ktrace = _debug_ktrace
Engine.bench_skinny = _bench_skinny
Engine.set_option = _set_option


def device_info(device_id: int = 0) -> dict:
    """name / total and free memory / HIP runtime version of a device (what asr.py:501-506 reads from torch.cuda)."""
    lib = load_library()
    name = C.create_string_buffer(256)
    tot, fr, ver = C.c_int64(0), C.c_int64(0), C.c_int32(0)
    if lib.sonic_device_info(int(device_id), name, 256, C.byref(tot), C.byref(fr), C.byref(ver)) != 0:
        raise RuntimeError((lib.sonic_last_error(None) or b"").decode())
    return {"name": name.value.decode(), "total_bytes": int(tot.value), "free_bytes": int(fr.value), "hip_runtime_version": int(ver.value)}


def runtime_info(device_id: int = 0) -> dict:
    """Hardware queues the HIP runtime of this process really has on the device (measured), GPU_MAX_HW_QUEUES as it reads now, and what an
    engine with slots wants (sonic_runtime_info)."""
    lib = load_library()
    q, env, want = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    if lib.sonic_runtime_info(int(device_id), C.byref(q), C.byref(env), C.byref(want)) != 0:
        raise RuntimeError((lib.sonic_last_error(None) or b"").decode())
    return {"hw_queues": int(q.value), "hw_queues_env": int(env.value), "hw_queues_wanted": int(want.value)}


def device_count() -> int:
    return int(load_library().sonic_device_count())
