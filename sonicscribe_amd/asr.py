"""Drop-in ``ASRModel`` (reference: backend/asr.py) backed by the MI355X HIP engine.

Same constructor, ``transcribe`` signature, return types, error behaviour and auxiliary members
(``.model``, ``get_model_info``) as the reference class, so ``backend/models_manager.py:32`` only has to
import this class instead of ``asr.ASRModel``.  What changed underneath:

  * asr.py:230-278  temp-WAV round trip          -> frontend.normalise_to_int16 (no disk)
  * asr.py:393-399  HF processor feature step    -> log-mel HIP kernel (csrc/logmel.hip)
  * asr.py:407-422  HF model.generate            -> HIP encoder / prefill / hipGraph greedy loop
  * asr.py:425-429  batch_decode                 -> unchanged (tokenizer stays in Python)

Concurrent callers (3 executor threads + the event loop, main.py:429-430, transcription_manager.py:58) are
batched per device by ``dispatch.Dispatcher`` instead of serialising on the model; with ``device="cuda:*"`` one engine
replica runs on every visible MI355X inside this one process (the reference is a single process, main.py:1001).
``submit()`` / ``transcribe_async()`` return without blocking the caller, so the WebSocket event loop can keep all
sessions' decodes in flight (INTEGRATION.md shows the change in transcription_manager.py).
"""
from __future__ import annotations

import asyncio
import time
from concurrent.futures import Future
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Union

import numpy as np

from . import frontend
from .dispatch import Dispatcher
from .engine import Engine, MODE_INT8, MODE_NATIVE, SonicError, device_count, device_info
from .spec import FULL, ModelDims


# Batches in flight per replica on ONE weight copy (engine slots, include/sonic_hip.h sonic_slot_create).  The reference's file mode keeps up to
# three decodes in flight on its one model object (backend/main.py:429-445); here they overlap on the device instead of serialising.
DEFAULT_SLOTS = 2
# Row-level scheduling (dispatch._ContinuousReplica): the replica's engine decodes forever over its rows, the slot prefills; requests join and
# leave row by row.  False: batch by batch (dispatch._Replica), every slot runs whole batches.
DEFAULT_CONTINUOUS = True


# --------------------------------------------------------------------------------------- prompts
class SyntheticPrompt:
    """Stand-in for the checkpoint's tokenizer + chat template (absent offline): fixed prefix / suffix ids
    around the audio placeholders.  Decoding renders ids as text so the call surface stays str-valued."""

    def __init__(self, dims: ModelDims, prefix: Sequence[int] = (1, 17, 23, 5), suffix: Sequence[int] = (7, 301, 302, 303, 9, 11)):
        self.dims, self.prefix, self.suffix = dims, list(prefix), list(suffix)

    def build(self, instruction: str, n_audio: int) -> List[int]:
        extra = [] if instruction == frontend.BASE_INSTRUCTION else [2 + (sum(instruction.encode()) % 200)]
        return self.prefix + [self.dims.audio_token_id] * n_audio + self.suffix + extra

    def decode(self, ids: Sequence[int]) -> str:
        eos = set(self.dims.eos_ids)
        return " ".join(str(int(i)) for i in ids if int(i) not in eos)


class HFPrompt:
    """Tokenizer + chat template of a real checkpoint (processing_glmasr.py:178-180: the audio placeholder string is
    repeated ``num_audio_tokens`` times before tokenisation).  Prompts are cached per (instruction, n_audio) -- SURVEY §8f4."""

    def __init__(self, processor, dims: ModelDims):
        self.processor, self.dims = processor, dims
        self.audio_token = getattr(processor, "audio_token", "<|pad|>")
        self._cache: Dict[Any, List[int]] = {}

    def build(self, instruction: str, n_audio: int) -> List[int]:
        key = (instruction, n_audio)
        if key not in self._cache:
            messages = [{"role": "user", "content": [{"type": "audio", "url": ""}, {"type": "text", "text": instruction}]}]
            text = self.processor.tokenizer.apply_chat_template(messages, tokenize=False, add_generation_prompt=True,
                                                                chat_template=self.processor.chat_template)
            text = text.replace(self.audio_token, self.audio_token * n_audio, 1)
            self._cache[key] = list(self.processor.tokenizer(text, add_special_tokens=False)["input_ids"])
        return self._cache[key]

    def decode(self, ids: Sequence[int]) -> str:
        return self.processor.batch_decode([list(map(int, ids))], skip_special_tokens=True)[0]


def _text_future(inner: "Future", decode) -> "Future[str]":
    """Future of the transcript behind a dispatcher future of token ids.  Cancelling it (a session that went away) cancels the queued
    request as well, so it never reaches the device."""
    out: "Future[str]" = Future()

    def done(f):
        if out.done():
            return
        try:
            out.set_result(decode(f.result()).strip())
        except BaseException as ex:
            if not out.done():
                out.set_exception(ex)
    inner.add_done_callback(done)
    out.add_done_callback(lambda f: inner.cancel() if f.cancelled() else None)
    return out


class AudioStream:
    """Device-resident counterpart of the reference's per-connection chunk store.

    The reference keeps every 2048-byte WebSocket chunk in a host dict keyed by chunk id (backend/audio_manager.py:21-33, fed from
    backend/main.py:813-842), and for every partial / final decode concatenates a chunk range on the host
    (audio_manager.py:99-123), converts it to float (backend/transcription_manager.py:45-54) and hands the tensor to
    ASRModel.transcribe.  Here a chunk goes straight into a ring in HBM on the session's GPU; a decode names a chunk range and the
    int16 -> float -> peak-normalise -> PCM_16 steps run on the device (csrc/ingest.hip), bit-identical with the host path.
    """

    def __init__(self, model: "ASRModel", session: str, replica: int, buffer_seconds: float, margin_seconds: float = 10.0):
        self.model, self.session, self.replica = model, session, replica
        # A decode names a sample range and the range is only read when the replica reaches the request, so the ring is LARGER than the
        # buffer the session sees: chunks stay addressable for `buffer_seconds` (the reference's MAX_AUDIO_BUFFER_SECONDS, config.py:25),
        # and a queued request survives `margin_seconds` of further appends before the ring overwrites its oldest samples (the
        # reference concatenates on the host at call time and cannot lose audio that way).
        self.visible = int(buffer_seconds * model.target_sr)
        self.ring = model.models[replica].ring_create(int((buffer_seconds + margin_seconds) * model.target_sr))
        self._chunks: Dict[int, tuple] = {}       # chunk id -> (first sample index, samples)
        self.next_chunk_id = 0
        self._oldest = 0                          # smallest chunk id still in the buffer

    def add_audio_chunk(self, audio_data: bytes, timestamp: Optional[float] = None) -> int:
        """audio_manager.py:21-33: store one wire chunk (with its arrival time, data_basic.py:11-20), return its chunk id."""
        first = self.ring.append(audio_data)
        cid = self.next_chunk_id
        self.next_chunk_id += 1
        self._chunks[cid] = (first, len(audio_data) // 2, time.time() if timestamp is None else float(timestamp))
        floor = first + len(audio_data) // 2 - self.visible           # chunks older than the buffer (audio_manager.py:35-59 drops them by age)
        while self._oldest < cid and self._chunks[self._oldest][0] < floor:
            del self._chunks[self._oldest]
            self._oldest += 1
        return cid

    @property
    def oldest_chunk_id(self) -> int:
        return self._oldest

    def chunk_timestamp(self, chunk_id: int, default: float = 0.0) -> float:
        """arrival time of a chunk still in the buffer (AudioChunk.timestamp); `default` once it left"""
        c = self._chunks.get(int(chunk_id))
        return c[2] if c is not None else default

    def chunk_range_samples(self, start_chunk_id: int, end_chunk_id: int):
        """(first sample index, sample count) of chunks start..end inclusive, restricted to what the buffer still holds
        (audio_manager.py:76-79: ids that left the buffer are skipped)."""
        ids = [c for c in range(max(start_chunk_id, self._oldest), end_chunk_id + 1) if c in self._chunks]
        if not ids:
            raise ValueError(f"no audio left in the buffer for chunks {start_chunk_id}..{end_chunk_id}")
        for a, b in zip(ids, ids[1:]):
            if b != a + 1:
                raise ValueError("chunk range is not contiguous in the buffer")
        return self._chunks[ids[0]][0], sum(self._chunks[c][1] for c in ids)

    def submit_samples(self, first: int, n: int, max_new_tokens: int = 128, hotwords: Optional[List[str]] = None) -> "Future[str]":
        """Transcribe ring samples [first, first + n) (the >30 s split of connection_manager.py:206-214 cuts at byte offsets, not chunks)."""
        m = self.model
        windows = [self.ring.slice(first + s, e - s) for s, e in frontend.split_windows(n, m.dims)]
        n_audio, _ = frontend.request_audio_tokens(n, m.dims)
        prompt = m.prompt.build(frontend.build_instruction(hotwords), n_audio)
        inner = m._dispatcher.submit(windows, prompt, int(max_new_tokens), replica=self.replica)
        return _text_future(inner, m.prompt.decode)

    def submit_chunks(self, start_chunk_id: int, end_chunk_id: int, max_new_tokens: int = 128, hotwords: Optional[List[str]] = None) -> "Future[str]":
        """Transcribe chunks start..end inclusive (audio_manager.py:76-79 get_chunks_by_range + :115-123 concatenation)."""
        first, n = self.chunk_range_samples(start_chunk_id, end_chunk_id)
        return self.submit_samples(first, n, max_new_tokens, hotwords)

    async def transcribe_chunks(self, start_chunk_id: int, end_chunk_id: int, max_new_tokens: int = 128, hotwords: Optional[List[str]] = None) -> str:
        return await asyncio.wrap_future(self.submit_chunks(start_chunk_id, end_chunk_id, max_new_tokens, hotwords))

    def close(self):
        self.ring.close()
        self._chunks.clear()


# --------------------------------------------------------------------------------------- façade
class ASRModel:
    def __init__(self, checkpoint_dir: str, device: str = "cuda", mode: str = "native",
                 cpu_threads: Optional[int] = None, cpu_interop_threads: Optional[int] = None,
                 *, max_batch: int = 32, max_ctx: int = 1024, slots: int = DEFAULT_SLOTS, continuous: bool = DEFAULT_CONTINUOUS, decoders: int = 1, bulk: bool = False, native_dispatch: Optional[bool] = None, _dims: Optional[ModelDims] = None,
                 _synthetic_seed: Optional[int] = None, _allow_synthetic_prompt: bool = False, _options: Optional[Dict[str, int]] = None):
        if mode not in ["native", "int8"]:
            raise ValueError("mode must be either 'native' or 'int8'")            # asr.py:46-47
        dev = str(device)
        if dev.startswith("cpu"):
            raise RuntimeError("sonicscribe_amd runs on MI355X only: DEVICE=cpu has no HIP path (no CPU fallback by design)")
        n_dev = device_count()
        # "cuda" / "cuda:1" = one replica (the reference's surface); "cuda:*" = every visible GPU; "cuda:0,2,3" = those
        spec_ = dev.split(":", 1)[1] if ":" in dev else "0"
        self.device_indices = list(range(n_dev)) if spec_ in ("*", "all") else [int(x) for x in spec_.split(",") if x != ""]
        if not self.device_indices or max(self.device_indices) >= n_dev:
            raise RuntimeError(f"HIP device {spec_} not available ({n_dev} visible)")
        self.device_index = self.device_indices[0]
        self.device = f"cuda:{self.device_index}" if len(self.device_indices) == 1 else "cuda:" + ",".join(map(str, self.device_indices))
        self.mode = mode
        self.model_dtype = "bfloat16" if mode == "native" else "float16"                     # asr.py:61
        emode = MODE_NATIVE if mode == "native" else MODE_INT8
        self.checkpoint_dir = Path(checkpoint_dir)
        self.target_sr = 16000
        self.is_glm_asr = True
        self.processor = None
        self.models: List[Engine] = []
        if _synthetic_seed is not None:
            self.dims = _dims or FULL
            for di in self.device_indices:
                eng = Engine(self.dims, di, emode, max_batch, max_ctx)
                eng.load_synthetic(_synthetic_seed)
                self.models.append(eng)
            self.prompt = SyntheticPrompt(self.dims)
        else:
            from . import weights
            self.dims = weights.load_dims(str(self.checkpoint_dir))
            for di in self.device_indices:
                eng = Engine(self.dims, di, emode, max_batch, max_ctx)
                weights.load_checkpoint(eng, str(self.checkpoint_dir))
                self.models.append(eng)
            try:
                from transformers import AutoProcessor
                self.processor = AutoProcessor.from_pretrained(str(self.checkpoint_dir))
                self.target_sr = self.processor.feature_extractor.sampling_rate
                self.prompt = HFPrompt(self.processor, self.dims)
            except Exception as ex:
                # The reference fails loudly when the processor / tokenizer is missing (asr.py:66, 120-146).  Feeding placeholder
                # prompt ids to a real model would return garbage "transcripts" without an error, so the stand-in prompt is
                # strictly opt-in (loader tests on tokenizer-less synthetic checkpoints).
                if not _allow_synthetic_prompt:
                    raise RuntimeError(f"could not load the processor / tokenizer from {self.checkpoint_dir}: {ex}") from ex
                self.prompt = SyntheticPrompt(self.dims)
        for eng in self.models:                      # experiment knobs (sonic_set_option) before the slots copy them
            for k, v in (_options or {}).items():
                eng.set_option(k, int(v))
        self.model = self.models[0]                  # main.py:84-86 checks and deletes `.model`
        # continuous: `decoders` handles per replica run a greedy loop over max_batch rows each, the other handles prefill (>= 1).  Streaming:
        # decoders=1, slots=2.  Bulk transcription of many segments: max_batch=64, decoders=3, slots=4 (the bench's pipeline shape since round 5; decoders=2, slots=3 before).
        # bulk=True (file mode: deep queues of whole segments): the same handles behind the library's native pipeline (dispatch._BulkReplica ->
        # csrc/pipeline.cpp); implies continuous decode loops.  max_batch=64, decoders=3, slots=4 is the bench's shape.
        self.bulk = bool(bulk)
        self.continuous = bool(continuous) or self.bulk
        self.decoders = max(1, int(decoders)) if self.continuous else 0
        self.slots = max(self.decoders + 1 if self.continuous else 1, int(slots))
        self._slot_engines = [[eng.slot() for _ in range(self.slots - 1)] for eng in self.models]     # same weights, further batches in flight
        self._dispatcher = Dispatcher(self.models, slots=self._slot_engines, continuous=self.continuous, decoders=self.decoders or 1,
                                      adaptive_tiles="gemm_small_eff" not in (_options or {}), bulk=self.bulk, native=native_dispatch)
        print(f"🚀 初始化 ASR 模型 | 模式: {mode.upper()} | 设备: {self.device} (MI355X HIP engine, "
              f"{self.model.weight_bytes() / 2**20:.0f} MiB weights x {len(self.models)} replica(s), {self.slots} batch slot(s) each)")

    @classmethod
    def from_synthetic(cls, dims: ModelDims = FULL, seed: int = 20260128, device: str = "cuda", mode: str = "native", **kw) -> "ASRModel":
        return cls("<synthetic>", device=device, mode=mode, _dims=dims, _synthetic_seed=seed, **kw)

    # -- reference helpers kept under their reference names
    def _format_hotwords_prompt(self, hotwords: List[str], max_hotwords: int = 10) -> str:
        return frontend.format_hotwords_prompt(hotwords, max_hotwords)

    def _prepare(self, audio_tensor, sampling_rate: int):
        wav = audio_tensor
        if hasattr(wav, "detach"):
            wav = wav.detach().cpu().numpy()
        wav = np.asarray(wav, dtype=np.float32)
        if wav.ndim == 2:
            wav = wav[0]                                                      # first channel (asr.py:252)
        if sampling_rate != self.target_sr:
            wav = frontend.resample_sinc_hann(wav, sampling_rate, self.target_sr)   # asr.py:255-261
        pcm = frontend.normalise_to_int16(wav)
        wins = frontend.split_windows(len(pcm), self.dims)
        n_audio, _ = frontend.request_audio_tokens(len(pcm), self.dims)
        return pcm, [pcm[s:e] for s, e in wins], n_audio

    def submit(self, audio_tensor, sampling_rate: int = 16000, max_new_tokens: int = 128, hotwords: Optional[List[str]] = None,
               session: Optional[str] = None) -> "Future[str]":
        """Non-blocking form of transcribe(): queues the request on a replica and returns a Future of the transcript.  `session`
        (e.g. the WebSocket client id) keeps a session's decodes on one GPU."""
        if not hasattr(self, "model"):
            raise RuntimeError("ASR model has been released")
        pcm, windows, n_audio = self._prepare(audio_tensor, sampling_rate)
        prompt = self.prompt.build(frontend.build_instruction(hotwords), n_audio)
        inner = self._dispatcher.submit(windows, prompt, int(max_new_tokens), session=session)
        return _text_future(inner, self.prompt.decode)

    def open_stream(self, session: str, buffer_seconds: float = 30.0, margin_seconds: float = 10.0) -> AudioStream:
        """A streaming session whose audio stays on the device (config.py:25 MAX_AUDIO_BUFFER_SECONDS = 30): chunks are appended to a
        ring on the session's GPU, partial / final decodes name chunk ranges (AudioStream).  The ring holds `margin_seconds` more than
        the buffer, so a max-length final that waits in the queue is not overwritten by the chunks that keep arriving."""
        if not hasattr(self, "model"):
            raise RuntimeError("ASR model has been released")
        return AudioStream(self, session, self._dispatcher.home(session), buffer_seconds, margin_seconds)

    async def transcribe_async(self, audio_tensor, sampling_rate: int = 16000, max_new_tokens: int = 128,
                               hotwords: Optional[List[str]] = None, session: Optional[str] = None) -> str:
        """Awaitable transcribe() for the asyncio callers (connection_manager.py:127-245): the event loop is not blocked while the
        device works, so all sessions' partial and final decodes can be in flight (and batched) together."""
        return await asyncio.wrap_future(self.submit(audio_tensor, sampling_rate, max_new_tokens, hotwords, session))

    def transcribe(self, audio_tensor, sampling_rate: int = 16000, max_new_tokens: int = 128,
                   hotwords: Optional[List[str]] = None, return_debug_info: bool = False) -> Union[str, Dict[str, Any]]:
        if not hasattr(self, "model"):
            raise RuntimeError("ASR model has been released")
        t0 = time.time()
        try:
            transcript = self.submit(audio_tensor, sampling_rate, max_new_tokens, hotwords).result()
            elapsed = time.time() - t0
            if return_debug_info:
                n = audio_tensor.shape[-1] if hasattr(audio_tensor, "shape") else len(audio_tensor)
                alloc, reserved = self.model.memory_info()           # asr.py:453-457: allocator state, not the weight size
                return {"transcript": transcript, "processing_time": elapsed, "audio_length_sec": n / sampling_rate,
                        "mode": self.mode, "device": str(self.device),
                        "gpu_memory_allocated_mb": alloc / 1024 ** 2, "gpu_memory_reserved_mb": reserved / 1024 ** 2}
            return transcript
        except RuntimeError as e:
            if "out of memory" in str(e).lower():
                print("⚠️ 显存不足！建议：使用更短的音频 / 减少 max_new_tokens")
            raise
        except Exception as e:
            print(f"❌ 转录过程中发生错误: {e}")
            raise

    def transcribe_batch(self, audios: Sequence[Any], sampling_rate: int = 16000, max_new_tokens: Union[int, Sequence[int]] = 128,
                         hotwords: Optional[List[str]] = None) -> List[str]:
        """Batched extension (the reference is B=1 per call): one device batch, per-segment results identical to transcribe()."""
        mn = [int(max_new_tokens)] * len(audios) if isinstance(max_new_tokens, int) else [int(x) for x in max_new_tokens]
        segs, req_win, prompts = [], [0], []
        instruction = frontend.build_instruction(hotwords)
        for a in audios:
            _, wins, n_audio = self._prepare(a, sampling_rate)
            segs.extend(wins)
            req_win.append(len(segs))
            prompts.append(self.prompt.build(instruction, n_audio))
        if len(self.models) == 1 and not self.continuous:
            ids, _ = self.model.transcribe_batch(segs, prompts, mn, req_win=req_win)
        else:            # independent segments: spread over the replicas (least-loaded placement), results in input order
            futs = [self._dispatcher.submit(segs[req_win[i]:req_win[i + 1]], prompts[i], mn[i]) for i in range(len(audios))]
            ids = [f.result() for f in futs]
        return [self.prompt.decode(i).strip() for i in ids]

    def get_model_info(self) -> Dict[str, Any]:
        """asr.py:490-513: the reference's keys for a GPU device (`cuda_version` carries the HIP runtime version: torch.version.cuda is
        the toolkit the reference's torch was built with), plus `engine`, `replicas`, `weights_mb`."""
        info = {"mode": self.mode, "device": str(self.device), "model_dtype": "torch." + self.model_dtype, "target_sampling_rate": self.target_sr,
                "checkpoint_dir": str(self.checkpoint_dir), "is_glm_asr": self.is_glm_asr}
        di = device_info(self.device_index)
        v = di["hip_runtime_version"]
        info.update({"cuda_version": f"HIP {v // 10000000}.{(v // 100000) % 100}.{v % 100000}", "gpu_name": di["name"],
                     "gpu_memory_total_mb": di["total_bytes"] / 1024 ** 2})
        info.update({"engine": "sonicscribe_amd/gfx950", "replicas": len(self.__dict__.get("models", [])), "slots_per_replica": self.__dict__.get("slots", 1), "continuous": self.__dict__.get("continuous", False), "bulk": self.__dict__.get("bulk", False),
                     "weights_mb": self.model.weight_bytes() / 1024 ** 2 if hasattr(self, "model") else 0.0})
        return info

    def close(self):
        c = self.__dict__.pop("_dispatcher", None)
        if c is not None:
            c.close()
        self.__dict__.pop("model", None)
        self.__dict__.pop("_slot_engines", None)
        for m in self.__dict__.pop("models", []):
            m.close()                                # (an engine closes its slots first)

    def __delattr__(self, name):   # main.py:84-86 does `del asr_model.model`
        if name == "model":
            self.close()
        else:
            super().__delattr__(name)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
