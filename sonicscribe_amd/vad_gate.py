"""Batched VAD gate: the per-session state machine of backend/vad_processor_manager.py::process_vad (:42-182),
restated for N sessions per call on numpy arrays.

The reference runs one `VADProcessorManager` per WebSocket session inside that session's asyncio `vad_loop`
(connection_manager.py:43-106): every tick it looks at the newest chunks of the session's buffer, accumulates 10 x 64 ms chunks,
asks Silero (`vad.py:84-126`) whether the 640 ms window holds speech at the session's CURRENT threshold, and updates hysteresis
counters and a dynamic threshold (0.3 -> 0.9).  At 128 sessions that is 128 Python state machines and 128 single-window network
calls per 64 ms.  Here the state of all sessions lives in flat arrays and one tick is two vectorised calls:

    ready, windows, thr = gate.offer(latest_ids, n_latest)     # accumulate; which sessions have a 10-chunk window, its chunk ids, thresholds
    is_speech = <the VAD network on the ready windows, at thr>  # stays outside: Silero's weights are not available offline (unpinned)
    changed, start_id, end_id = gate.decide(ready, is_speech)   # counters, threshold, speaking state; -1 = None

`tick_scores()` folds both for a VAD that yields one score per window (speech iff score > threshold).

Pinned bit-exactly (thresholds as float64, every tick, every field) against fixtures produced by the reference file itself under a
scripted VAD: tests/golden/vad_gate.npz, oracle/gen_vad_fixtures.py, tests/test_vad_gate.py.  Reference behaviours kept on purpose:
  * `get_chunks_for_vad` (audio_manager.py:60-68) returns only the newest VAD_SMOOTHING_WINDOW = 2 chunks (nothing ever marks a
    chunk processed), so bursts skip chunks and a chunk can re-enter the accumulator after its window was consumed;
  * one window per call at most; `speech_end_id` is the LAST id of the whole accumulator (which can hold an 11th chunk), the start id
    its first (vad_processor_manager.py:127-129,156);
  * the four-way threshold rule (:122-167) with plain Python float arithmetic = IEEE double, clamped to [min, max] every window.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import numpy as np


@dataclass(frozen=True)
class GateConfig:
    """backend/config.py:28-37"""
    window: int = 10             # VAD_PROCESS_WINDOW
    smoothing: int = 2           # VAD_SMOOTHING_WINDOW (also the `get_chunks_for_vad` look-back)
    thr_init: float = 0.3        # VAD_INITIAL_THRESHOLD
    thr_min: float = 0.3         # VAD_THRESHOLD_MIN
    thr_max: float = 0.9         # VAD_THRESHOLD_MAX
    thr_step: float = 0.1        # VAD_THRESHOLD_STEP


class BatchedVADGate:
    def __init__(self, n_sessions: int, cfg: GateConfig = GateConfig()):
        self.n, self.cfg = int(n_sessions), cfg
        n, cap = self.n, cfg.window + cfg.smoothing + 1
        self.acc = np.full((n, cap), -1, np.int64)            # chunk_accumulator (ids, in insertion order until a window sorts them)
        self.acc_len = np.zeros(n, np.int64)
        self.speaking = np.zeros(n, bool)                     # vad_is_speaking
        self.speech_count = np.zeros(n, np.int64)
        self.silence_count = np.zeros(n, np.int64)
        self.threshold = np.full(n, cfg.thr_init, np.float64)  # current_vad_threshold
        self.speech_start_chunk_id = np.full(n, -1, np.int64)
        self.last_processed_chunk_id = np.full(n, -1, np.int64)
        self._pend_first = np.full(n, -1, np.int64)           # of the window handed out by offer(): accumulator[0] and accumulator[-1]
        self._pend_last = np.full(n, -1, np.int64)

    def reset(self, idx) -> None:
        """A session (re)connects: a fresh VADProcessorManager (connection_manager.py:27)."""
        c = self.cfg
        self.acc[idx] = -1; self.acc_len[idx] = 0; self.speaking[idx] = False; self.speech_count[idx] = 0; self.silence_count[idx] = 0
        self.threshold[idx] = c.thr_init; self.speech_start_chunk_id[idx] = -1; self.last_processed_chunk_id[idx] = -1

    # ---- step 1: accumulate ----------------------------------------------------------------------------------------------------------
    def offer(self, latest_ids: np.ndarray, n_latest: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """latest_ids [N][smoothing]: the newest chunk ids of every session's buffer in ascending order (what `get_chunks_for_vad`
        returns), n_latest [N] how many of them are valid (0: the buffer is empty -> the session's call returns early, :55-57).
        Returns (ready [N] bool, windows [N][window] chunk ids of the window to classify (-1 where not ready), threshold [N])."""
        c, n = self.cfg, self.n
        latest_ids = np.asarray(latest_ids, np.int64).reshape(n, -1)
        n_latest = np.asarray(n_latest, np.int64)
        for j in range(latest_ids.shape[1]):                  # `for chunk in recent_chunks: if chunk.chunk_id not in accumulator: append` (:64-66)
            cid = latest_ids[:, j]
            valid = j < n_latest
            self.last_processed_chunk_id = np.where(valid & (cid > self.last_processed_chunk_id), cid, self.last_processed_chunk_id)   # (:60-61)
            present = ((self.acc == cid[:, None]) & (np.arange(self.acc.shape[1])[None, :] < self.acc_len[:, None])).any(axis=1)
            add = valid & ~present
            rows = np.nonzero(add)[0]
            self.acc[rows, self.acc_len[rows]] = cid[rows]
            self.acc_len[rows] += 1
        ready = (n_latest > 0) & (self.acc_len >= c.window)    # (:69-71)
        windows = np.full((n, c.window), -1, np.int64)
        rows = np.nonzero(ready)[0]
        if rows.size:
            a = self.acc[rows].copy()
            a[np.arange(a.shape[1])[None, :] >= self.acc_len[rows, None]] = np.iinfo(np.int64).max
            a.sort(axis=1)                                      # `chunk_accumulator.sort(key=chunk_id)` (:74)
            keep = np.arange(a.shape[1])[None, :] < self.acc_len[rows, None]
            self.acc[rows] = np.where(keep, a, -1)
            windows[rows] = a[:, :c.window]                     # `chunk_accumulator[:processing_window]` (:85-86)
            self._pend_first[rows] = a[:, 0]
            self._pend_last[rows] = a[np.arange(rows.size), self.acc_len[rows] - 1]
        return ready, windows, self.threshold.copy()

    # ---- step 2: the window's verdict ------------------------------------------------------------------------------------------------
    def decide(self, ready: np.ndarray, is_speech: np.ndarray, valid: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """is_speech [N] for the sessions `offer` reported ready (ignored elsewhere).  `valid` False = the window held no samples: it is
        dropped without touching the state (:90-94).  Returns (state_changed [N] bool, speech_start_id [N], speech_end_id [N]; -1 = None)."""
        c, n = self.cfg, self.n
        ready = np.asarray(ready, bool)
        sp = np.asarray(is_speech, bool)
        live = ready if valid is None else (ready & np.asarray(valid, bool))
        S = c.smoothing
        up, down = live & sp, live & ~sp
        # hysteresis counters (:107-114)
        self.speech_count = np.where(up, np.minimum(self.speech_count + 1, S), np.where(down, np.maximum(0, self.speech_count - 1), self.speech_count))
        self.silence_count = np.where(up, np.maximum(0, self.silence_count - 1), np.where(down, np.minimum(self.silence_count + 1, S), self.silence_count))
        # the four cases, in the reference's if / elif order (:124-167)
        c1 = live & ~self.speaking & (self.speech_count >= 1)
        c2 = live & ~c1 & self.speaking & (self.speech_count > 0)
        c3 = live & ~c1 & ~c2 & self.speaking & (self.silence_count >= S)
        c4 = live & ~c1 & ~c2 & ~c3 & ~self.speaking & (self.silence_count >= S)
        thr = self.threshold
        thr = np.where(c1, np.minimum(thr + c.thr_step, c.thr_max), thr)
        thr = np.where(c2, np.minimum(thr + c.thr_step * 0.3, c.thr_max), thr)
        thr = np.where(c3 | c4, c.thr_min, thr)
        thr = np.where(live, np.maximum(c.thr_min, np.minimum(c.thr_max, thr)), thr)     # boundary protection (:171)
        self.threshold = thr
        start_id = np.where(c1, self._pend_first, -1)
        end_id = np.where(c3, self._pend_last, -1)
        self.speech_start_chunk_id = np.where(c1, self._pend_first, self.speech_start_chunk_id)
        self.speaking = np.where(c1, True, np.where(c3, False, self.speaking))
        # the consumed window leaves the accumulator (:174; also for an empty window, :92)
        rows = np.nonzero(ready)[0]
        if rows.size:
            w = c.window
            self.acc[rows, :-w] = self.acc[rows, w:]
            self.acc[rows, -w:] = -1
            self.acc_len[rows] -= w
        return c1 | c3, start_id, end_id

    def tick_scores(self, latest_ids, n_latest, score_of_window: Callable[[np.ndarray, np.ndarray], np.ndarray]):
        """One tick for a VAD that maps a window to a score: speech iff score > threshold (vad.py:84-126 with the session's dynamic
        threshold).  score_of_window(rows, windows[rows]) -> float array."""
        ready, windows, thr = self.offer(latest_ids, n_latest)
        rows = np.nonzero(ready)[0]
        sp = np.zeros(self.n, bool)
        if rows.size:
            sp[rows] = np.asarray(score_of_window(rows, windows[rows]), np.float64) > thr[rows]
        return (ready, windows) + self.decide(ready, sp)


def newest_chunks(next_chunk_id: np.ndarray, oldest_chunk_id: np.ndarray, look_back: int = 2) -> Tuple[np.ndarray, np.ndarray]:
    """What `AudioBufferManager.get_chunks_for_vad()` (audio_manager.py:60-68) returns for buffers holding chunk ids
    [oldest, next): the newest `look_back` ids in ascending order, left-aligned, and how many there are."""
    nxt, old = np.asarray(next_chunk_id, np.int64), np.asarray(oldest_chunk_id, np.int64)
    cnt = np.clip(nxt - old, 0, look_back)
    ids = (nxt - cnt)[:, None] + np.arange(look_back)[None, :]
    return np.where(np.arange(look_back)[None, :] < cnt[:, None], ids, -1), cnt
