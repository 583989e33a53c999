"""SURVEY.md §8 f3 (the pinnable half): the batched VAD gate state machine against fixtures produced by the reference's own
backend/vad_processor_manager.py::process_vad under a scripted VAD (oracle/gen_vad_fixtures.py), every tick, every field, bit-exact
(thresholds compared as float64 bit patterns).  The Silero network is not part of this: its weights are absent offline (unpinned)."""
import json
import os

import numpy as np

from sonicscribe_amd import frontend
from sonicscribe_amd.vad_gate import BatchedVADGate, GateConfig, newest_chunks

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _replay(batched: bool):
    g = np.load(os.path.join(GOLD, "vad_gate.npz"))
    arrivals, out_i, out_thr, score = g["arrivals"], g["out_i"], g["out_thr"], g["score_of_chunk"]
    k = json.loads(str(g["consts"]))
    cfg = GateConfig(window=k["window"], smoothing=k["smoothing"], thr_init=k["thr_init"], thr_min=k["thr_min"], thr_max=k["thr_max"], thr_step=k["thr_step"])
    assert cfg == GateConfig()                                  # the product's defaults ARE backend/config.py:28-37
    S, T = arrivals.shape
    groups = [np.arange(S)] if batched else [np.array([s]) for s in range(S)]
    n_changes = 0
    for rows in groups:
        gate = BatchedVADGate(len(rows))
        nxt = np.zeros(len(rows), np.int64)
        for t in range(T):
            nxt += arrivals[rows, t]
            ids, cnt = newest_chunks(nxt, np.zeros_like(nxt), cfg.smoothing)
            ready, windows, changed, st, en = gate.tick_scores(ids, cnt, lambda rr, w: score[rows[rr], w[:, 0]])
            want = out_i[rows, t]
            tag = np.where(ready, windows[:, 0] + 1, -1)        # the reference's window starts at this chunk (its first PCM sample is id + 1)
            got = np.stack([changed.astype(np.int64), st, en, gate.speaking.astype(np.int64), gate.speech_count, gate.silence_count, gate.acc_len, tag], axis=1)
            assert np.array_equal(got, want), (rows[np.nonzero((got != want).any(axis=1))[0][:3]], t, got[(got != want).any(axis=1)][:3], want[(got != want).any(axis=1)][:3])
            assert np.array_equal(gate.threshold.view(np.uint64), out_thr[rows, t].view(np.uint64)), (t, gate.threshold, out_thr[rows, t])
            n_changes += int(changed.sum())
    return n_changes


def test_gate_matches_reference_all_sessions_in_one_batch():
    assert _replay(batched=True) == 288                        # speech starts + ends in the fixture


def test_gate_matches_reference_one_session_at_a_time():
    assert _replay(batched=False) == 288


def test_gate_window_without_samples_and_reset():
    gate = BatchedVADGate(2)
    for t in range(12):
        ids, cnt = newest_chunks(np.array([t + 1, t + 1]), np.zeros(2, np.int64))
        ready, windows, thr = gate.offer(ids, cnt)
        if ready.any():
            # session 1's window decoded to zero samples: dropped without touching counters or threshold (vad_processor_manager.py:90-94)
            ch, st, en = gate.decide(ready, np.array([True, True]), valid=np.array([True, False]))
            assert ch.tolist() == [True, False] and st[0] == windows[0, 0] and st[1] == -1
            assert gate.speaking.tolist() == [True, False] and gate.threshold[1] == 0.3 and gate.threshold[0] == 0.4
            assert gate.acc_len.tolist() == [0, 0]
    gate.reset(0)
    assert not gate.speaking[0] and gate.threshold[0] == 0.3 and gate.acc_len[0] == 0


def test_hotwords_against_reference_fixture():
    """asr.py:303-333 called in the build container (oracle/gen_vad_fixtures.py): set() de-duplicates RAW strings, cleaning comes after."""
    cases = json.loads(str(np.load(os.path.join(GOLD, "hotwords.npz"))["cases"]))
    prefix = ". Pay special attention to these important terms: "
    for c in cases:
        s = frontend.format_hotwords_prompt(c["input"])
        got = sorted(s[len(prefix):].split(", ")) if s else []
        assert (s == "") == (c["n_entries"] == 0) and (s == "" or s.startswith(prefix))
        assert len(got) == c["n_entries"], (c["input"], got)
        if c["n_entries"] < 10:                                 # at the cap the reference keeps a set()-order-dependent subset
            assert got == c["sorted_entries"], (c["input"], got, c["sorted_entries"])
        else:
            cleaned = {f'"{h.strip().lower()}"' for h in c["input"] if isinstance(h, str) and h.strip()}
            assert set(got) <= cleaned and set(c["sorted_entries"]) <= cleaned
