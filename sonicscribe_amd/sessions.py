"""N streaming sessions behind one gate: the vectorised counterpart of N `ConnectionManager.vad_loop`s
(backend/connection_manager.py:43-106) over device-resident audio.

Per session the reference runs, every >= 64 ms: `process_vad()` (vad_processor_manager.py:42-182); on a speech start
`create_speech_segment` (audio_manager.py:81-95); while speaking a partial decode of the newest <= 20 chunks at most once a second
(connection_manager.py:89-92,127-166, audio_manager.py:106-114, 15 new tokens: transcription_manager.py:19-28); on a speech end
`finalize_current_segment` + a final decode of the segment's chunks up to the newest one (audio_manager.py:115-123), split into
<= 30 s pieces at byte offsets when longer (connection_manager.py:169-245), `min(50 + int(5 * seconds), 200)` new tokens each
(transcription_manager.py:30-41).  Here one `tick()` does that for all sessions: one batched gate step (vad_gate.BatchedVADGate),
and every decode it triggers is a chunk / sample range of the session's ring in HBM handed to the coalescing dispatcher
(asr.AudioStream.submit_chunks) - the requests of one tick end up in one device batch per step class.

The VAD network is a caller-supplied function: `vad(rows, windows_pcm, thresholds) -> bool[len(rows)]` gets, for every session with
a complete 640 ms window, that window's int16 samples and the session's current dynamic threshold (Silero stays outside, as in the
reference: its weights are not available offline).
"""
from __future__ import annotations

import time
from collections import deque
from concurrent.futures import Future
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from .vad_gate import BatchedVADGate, GateConfig, newest_chunks

CHUNK_BYTES = 2048                   # config.py:24 AUDIO_CHUNK_SIZE (64 ms of 16 kHz int16)
PARTIAL_CHUNKS = 20                  # config.py:40 TEMPORARY_TRANSCRIPTION_INTERVAL
PARTIAL_TOKENS = 15                  # transcription_manager.py:24
PARTIAL_PERIOD_S = 1.0               # connection_manager.py:89-92
MAX_SEGMENT_S = 30.0                 # config.py:41 MAX_SEGMENT_DURATION


def committed_max_new_tokens(seconds: float) -> int:
    return min(50 + int(seconds * 5), 200)          # transcription_manager.py:37


class GatedSessions:
    def __init__(self, model, session_ids: Sequence[str], buffer_seconds: float = 30.0, hotwords: Optional[List[str]] = None,
                 cfg: GateConfig = GateConfig()):
        self.model, self.ids, self.hotwords, self.cfg = model, list(session_ids), hotwords, cfg
        n = len(self.ids)
        self.streams = [model.open_stream(s, buffer_seconds) for s in self.ids]
        self.gate = BatchedVADGate(n, cfg)
        self.recent: List[deque] = [deque(maxlen=cfg.window + cfg.smoothing + 2) for _ in range(n)]   # (chunk id, bytes): what the VAD needs
        self.segment_start = np.full(n, -1, np.int64)     # current_segment.start_chunk_id (-1: none)
        self.last_partial = np.zeros(n, np.float64)
        self.sr = model.target_sr

    def add_audio_chunk(self, s: int, audio_data: bytes) -> int:
        """main.py:813-842 -> connection_manager.py:108-125: one wire chunk of session s."""
        cid = self.streams[s].add_audio_chunk(audio_data)
        self.recent[s].append((cid, audio_data))
        return cid

    def _window_pcm(self, s: int, ids: np.ndarray) -> np.ndarray:
        have = dict(self.recent[s])
        return np.frombuffer(b"".join(have.get(int(c), b"") for c in ids), dtype=np.int16)

    def tick(self, vad: Callable[[np.ndarray, List[np.ndarray], np.ndarray], np.ndarray], now: Optional[float] = None) -> List[Dict]:
        """One pass of every session's vad_loop body.  Returns the events of this tick: dicts with `session`, `type` in
        {"speech_start", "partial", "final"}, chunk ids and, for decodes, `future` (a Future of the transcript)."""
        now = time.time() if now is None else now
        n = len(self.ids)
        nxt = np.array([st.next_chunk_id for st in self.streams], np.int64)
        old = np.array([st.oldest_chunk_id for st in self.streams], np.int64)
        ids, cnt = newest_chunks(nxt, old, self.cfg.smoothing)
        ready, windows, thr = self.gate.offer(ids, cnt)
        rows = np.nonzero(ready)[0]
        sp, valid = np.zeros(n, bool), np.ones(n, bool)
        if rows.size:
            pcm = [self._window_pcm(int(r), windows[r]) for r in rows]
            valid[rows] = [len(p) > 0 for p in pcm]
            sp[rows] = np.asarray(vad(rows, pcm, thr[rows]), bool)
        changed, start_id, end_id = self.gate.decide(ready, sp, valid)
        events: List[Dict] = []
        for s in np.nonzero(changed)[0]:
            s = int(s)
            st = self.streams[s]
            if start_id[s] >= 0:                                          # create_speech_segment (connection_manager.py:66-72)
                self.segment_start[s] = start_id[s]
                events.append({"session": self.ids[s], "type": "speech_start", "start_chunk_id": int(start_id[s])})
            if end_id[s] >= 0 and self.segment_start[s] >= 0:             # finalize_current_segment + committed transcription (:74-84)
                seg0, self.segment_start[s] = int(self.segment_start[s]), -1
                events.extend(self._final(s, st, seg0, int(end_id[s])))
        speaking = self.gate.speaking
        for s in np.nonzero(speaking & (self.segment_start >= 0) & (now - self.last_partial >= PARTIAL_PERIOD_S))[0]:
            s = int(s)
            st = self.streams[s]
            lo, hi = max(int(self.segment_start[s]), st.next_chunk_id - PARTIAL_CHUNKS), st.next_chunk_id - 1   # audio_manager.py:106-114
            self.last_partial[s] = now
            try:
                first, ns = st.chunk_range_samples(lo, hi)
            except ValueError:
                continue
            if ns * 2 < CHUNK_BYTES:                                      # transcription_manager.py:21-22
                continue
            events.append({"session": self.ids[s], "type": "partial", "start_chunk_id": lo, "end_chunk_id": hi, "first_sample": first, "n_samples": ns,
                           "future": st.submit_samples(first, ns, PARTIAL_TOKENS, self.hotwords)})
        return events

    def _final(self, s: int, st, seg0: int, end_chunk_id: int) -> List[Dict]:
        try:
            first, ns = st.chunk_range_samples(seg0, st.next_chunk_id - 1)     # get_committed_audio_data: up to the NEWEST chunk (audio_manager.py:118)
        except ValueError:
            return []
        if ns * 2 < CHUNK_BYTES * 2:                                      # connection_manager.py:175-177
            return []
        out = []
        piece = int(MAX_SEGMENT_S * self.sr)
        n_sub = -(-ns // piece)
        for i in range(n_sub):                                            # <= 30 s: one request; longer: pieces cut at sample 480000 * i
            a, b = i * piece, min(ns, (i + 1) * piece)
            if (b - a) * 2 < CHUNK_BYTES * 2:                               # transcribe_committed returns "" below two chunks (:32-33)
                continue
            out.append({"session": self.ids[s], "type": "final", "start_chunk_id": seg0, "end_chunk_id": end_chunk_id, "part": i, "parts": n_sub,
                        "seconds": (b - a) / self.sr, "first_sample": first + a, "n_samples": b - a,
                        "future": st.submit_samples(first + a, b - a, committed_max_new_tokens((b - a) / self.sr), self.hotwords)})
        return out

    def close(self):
        for st in self.streams:
            st.close()
