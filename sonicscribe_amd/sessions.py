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
        # chunk id -> bytes: what the VAD windows are cut from.  The reference keeps the AudioChunk OBJECTS in chunk_accumulator
        # (vad_processor_manager.py:64-66,85-86), so a window always has its audio however slowly the loop ticks; here a chunk's bytes stay
        # until the gate's accumulator no longer names its id (pruned in tick()).
        self.recent: List[Dict[int, bytes]] = [dict() for _ in range(n)]
        self.segment_start = np.full(n, -1, np.int64)     # current_segment.start_chunk_id (-1: none)
        self.segment_start_time = np.zeros(n, np.float64)  # current_segment.start_time = arrival time of that chunk (connection_manager.py:69-72)
        self.last_partial = np.zeros(n, np.float64)
        self.sr = model.target_sr

    def add_audio_chunk(self, s: int, audio_data: bytes, timestamp: Optional[float] = None) -> int:
        """main.py:813-842 -> connection_manager.py:108-125: one wire chunk of session s (timestamp: its arrival time, data_basic.py:11-20)."""
        cid = self.streams[s].add_audio_chunk(audio_data, timestamp)
        self.recent[s][cid] = audio_data
        return cid

    def _window_pcm(self, s: int, ids: np.ndarray) -> np.ndarray:
        have = self.recent[s]
        return np.frombuffer(b"".join(have.get(int(c), b"") for c in ids), dtype=np.int16)

    def _prune_recent(self):
        """drop the bytes of chunks that neither the gate's accumulator names nor the next look-back can offer again"""
        g = self.gate
        for s, have in enumerate(self.recent):
            floor = self.streams[s].next_chunk_id - self.cfg.smoothing
            if g.acc_len[s] > 0:
                floor = min(floor, int(g.acc[s, :g.acc_len[s]].min()))
            for c in [c for c in have if c < floor]:
                del have[c]

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
        self._prune_recent()
        events: List[Dict] = []
        for s in np.nonzero(changed)[0]:
            s = int(s)
            st = self.streams[s]
            if start_id[s] >= 0:                                          # create_speech_segment (connection_manager.py:66-72)
                self.segment_start[s] = start_id[s]
                self.segment_start_time[s] = st.chunk_timestamp(int(start_id[s]), now)
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
        # segment_duration = min(actual_duration, segment.duration): the timestamp span start chunk -> speech_end_id chunk bounds both the
        # split test and the unsplit request's token budget (connection_manager.py:186-192); the audio itself runs to the NEWEST chunk
        seg_dur = min(ns / self.sr, max(0.0, st.chunk_timestamp(end_chunk_id, time.time()) - float(self.segment_start_time[s])))
        if seg_dur <= MAX_SEGMENT_S:                                      # one request, whatever its audio length (the processor windows it)
            return [{"session": self.ids[s], "type": "final", "start_chunk_id": seg0, "end_chunk_id": end_chunk_id, "part": 0, "parts": 1,
                     "seconds": seg_dur, "first_sample": first, "n_samples": ns,
                     "future": st.submit_samples(first, ns, committed_max_new_tokens(seg_dur), self.hotwords)}]
        n_sub = -(-ns // piece)
        for i in range(n_sub):                                            # longer: pieces cut at sample 480000 * i, each with its own duration
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
