#!/usr/bin/env python3
"""Generate tests/golden/vad_gate.npz and tests/golden/hotwords.npz from the REFERENCE's own code.  TEST INFRASTRUCTURE ONLY.

Runs only in the build container (it imports files under /root/reference); never on the GPU box, never imported by the product.

  vad_gate.npz   backend/vad_processor_manager.py::VADProcessorManager.process_vad (:42-182) driven tick by tick, with the real
                 backend/audio_manager.py::AudioBufferManager and backend/config.py::AppConfig behind it.  Absent third-party
                 modules are stubbed: `dotenv.load_dotenv` (no .env file here), `silero_vad` / `torchaudio` (the Silero network
                 and its weights are not available offline) and `models_manager` (which would import backend/asr.py and with it
                 soundfile / torchaudio).  The stub VAD processor is SCRIPTED: is_voice_active(audio, threshold) returns
                 score > threshold, where the score is looked up by the first PCM sample of the 10-chunk window it is handed
                 (so the fixture also pins WHICH chunks the gate combines).  What the fixture pins is the per-session state
                 machine: accumulator, hysteresis counters, dynamic threshold, the (state_changed, start_id, end_id) outputs.
                 The Silero network itself stays unpinned (weights absent).
  hotwords.npz   backend/asr.py::ASRModel._format_hotwords_prompt (:303-333) called unbound (the method does not touch `self`) with
                 stubbed soundfile / torchaudio imports, for lists that exercise set() de-duplication of the RAW strings before
                 cleaning.  set() order varies between processes, so the fixture stores the SORTED entries of each result.

Usage:  python oracle/gen_vad_fixtures.py [--out tests/golden]
"""
from __future__ import annotations

import argparse
import asyncio
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/backend"


def _stub(name, **attrs):
    import importlib.machinery
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class ScriptedVAD:
    """Stands for vad.py::VADProcessor: `is_voice_active(audio, threshold)` (vad.py:84-126) -> score(window) > threshold."""

    def __init__(self):
        self.scores = {}          # tag (first sample of the window as int16) -> score
        self.calls = []           # (tag, n_samples, threshold)

    def is_voice_active(self, audio_tensor, threshold=None):
        tag = int(round(float(audio_tensor[0]) * 32768.0))
        self.calls.append((tag, int(audio_tensor.numel()), float(threshold)))
        return self.scores[tag] > threshold


def load_reference():
    _stub("dotenv", load_dotenv=lambda *a, **k: None)
    _stub("silero_vad", load_silero_vad=lambda: None, get_speech_timestamps=None, VADIterator=None, read_audio=None)
    _stub("torchaudio")
    vad = ScriptedVAD()
    _stub("models_manager", asr_model_get=lambda: None, vad_model_get=lambda: vad)
    sys.path.insert(0, REF)
    import vad_processor_manager as vpm          # the reference file itself
    import audio_manager as am
    import config as cfg
    return vpm, am, cfg, vad


def gen_vad(out_dir: str):
    vpm, am, cfg, vad = load_reference()
    rng = np.random.default_rng(20261003)
    n_sessions, n_ticks = 24, 420
    CH = cfg.AppConfig.AUDIO_CHUNK_SIZE // 2      # samples per 64 ms chunk
    arrivals = np.zeros((n_sessions, n_ticks), np.int8)
    # per tick: changed, start_id, end_id, speaking, speech_count, silence_count, acc_len, window tag consumed (-1: none)
    out_i = np.full((n_sessions, n_ticks, 8), -1, np.int32)
    out_thr = np.zeros((n_sessions, n_ticks), np.float64)
    score_of_chunk = np.zeros((n_sessions, n_ticks * 3 + 8), np.float64)
    for s in range(n_sessions):
        vad.scores.clear()
        buf = am.AudioBufferManager()
        gate = vpm.VADProcessorManager(buf)
        # scripted speech-likelihood per chunk: runs of speech / silence with noisy scores (looked up by the window's first chunk)
        n_chunks_max = score_of_chunk.shape[1]
        level, sc = 0.1, np.zeros(n_chunks_max)
        i = 0
        while i < n_chunks_max:
            run = int(rng.integers(8, 90))
            level = float(rng.choice([0.05, 0.2, 0.35, 0.5, 0.7, 0.85, 0.97]))
            sc[i:i + run] = np.clip(level + rng.normal(0, 0.12, size=min(run, n_chunks_max - i)), 0.0, 1.0)
            i += run
        score_of_chunk[s] = sc
        # arrival pattern: mostly one chunk per tick, sometimes none, sometimes bursts of 2-3 (a blocked event loop: chunks get skipped,
        # since the reference only ever looks at the latest VAD_SMOOTHING_WINDOW = 2 chunks of the buffer)
        pat = rng.choice([0, 1, 1, 1, 1, 1, 2, 3], size=n_ticks) if s % 3 else np.ones(n_ticks, np.int64)
        arrivals[s] = pat
        for t in range(n_ticks):
            for _ in range(int(pat[t])):
                cid = buf.next_chunk_id
                pcm = np.zeros(CH, np.int16)
                pcm[0] = cid + 1                                   # tag: chunk id + 1 (never 0, never > 32767 here)
                vad.scores[cid + 1] = float(sc[cid])
                buf.add_audio_chunk(pcm.tobytes())
            n_calls = len(vad.calls)
            changed, st, en = asyncio.run(gate.process_vad())
            tag = vad.calls[-1][0] if len(vad.calls) > n_calls else -1
            out_i[s, t] = [int(changed), -1 if st is None else st, -1 if en is None else en, int(gate.vad_is_speaking), gate.speech_count,
                           gate.silence_count, len(gate.chunk_accumulator), tag]
            out_thr[s, t] = gate.current_vad_threshold
    A = cfg.AppConfig
    consts = dict(window=A.VAD_PROCESS_WINDOW, smoothing=A.VAD_SMOOTHING_WINDOW, thr_init=A.VAD_INITIAL_THRESHOLD, thr_min=A.VAD_THRESHOLD_MIN,
                  thr_max=A.VAD_THRESHOLD_MAX, thr_step=A.VAD_THRESHOLD_STEP, chunk_bytes=A.AUDIO_CHUNK_SIZE, sample_rate=A.AUDIO_SAMPLE_RATE)
    np.savez_compressed(os.path.join(out_dir, "vad_gate.npz"), arrivals=arrivals, out_i=out_i, out_thr=out_thr, score_of_chunk=score_of_chunk,
                        consts=json.dumps(consts))
    print("vad_gate.npz:", n_sessions, "sessions x", n_ticks, "ticks;", int((out_i[:, :, 0] == 1).sum()), "state changes;", consts)


def gen_hotwords(out_dir: str):
    import transformers  # noqa: F401  (before the stubs below: its own availability probes must see the real environment)
    _stub("soundfile")
    _stub("torchaudio")
    sys.path.insert(0, REF)
    import asr as ref_asr                                          # the reference file itself (bitsandbytes absent: it only warns)
    fn = ref_asr.ASRModel._format_hotwords_prompt
    cases = [
        [], ["Alpha", "alpha "], ["Alpha", "Alpha", "beta"], ["Brand", "brand ", "Model X"], ["  ", None, 3], [" x ", "x", "X", "x"],
        [f"w{i}" for i in range(20)], ["Kubernetes", "grafana"], ["a", "", "b", "  c  ", None],
    ]
    outs = []
    for c in cases:
        s = fn(None, list(c))
        prefix = ". Pay special attention to these important terms: "
        entries = sorted(s[len(prefix):].split(", ")) if s else []
        outs.append({"input": [x if isinstance(x, (str, int)) else None for x in c], "n_entries": len(entries), "sorted_entries": entries,
                     "prefix_ok": (s == "" or s.startswith(prefix))})
    np.savez_compressed(os.path.join(out_dir, "hotwords.npz"), cases=json.dumps(outs))
    for o in outs:
        print("hotwords:", o["input"], "->", o["sorted_entries"])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    a = ap.parse_args()
    gen_hotwords(a.out)
    gen_vad(a.out)
