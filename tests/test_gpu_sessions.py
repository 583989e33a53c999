"""SURVEY.md §8 f3 on the device: 128 concurrent sessions through the batched VAD gate (sessions.GatedSessions) into ring decodes
(AudioStream.submit_samples -> dispatcher -> engine), every partial and final equal to `transcribe()` of the same wire bytes through the
host path; and a queued max-length request that survives further appends (the ring holds a margin beyond the visible buffer)."""
import threading

import numpy as np
import pytest

from sonicscribe_amd import frontend, spec, synth

pytestmark = pytest.mark.gpu

CHUNK = 1024     # samples per 64 ms wire chunk


def energy_vad(rows, pcm, thr):
    """stand-in for Silero (weights absent offline): a window is speech when its mean |s| / 12000 exceeds the session's dynamic threshold"""
    return np.array([np.abs(p.astype(np.int32)).mean() / 12000.0 > t for p, t in zip(pcm, thr)])


@pytest.mark.parametrize("dims_name", ["tiny", "full"])
def test_128_sessions_gate_to_ring_decodes(dims_name):
    """`full`: the same 128 sessions on GLM-ASR-Nano dimensions (32 + 28 layers, vocabulary 59264) - BASELINE config 5's session count on one GPU;
    a sample of the finals and partials is compared with solo `transcribe()` calls (every one at `tiny`)."""
    from sonicscribe_amd.asr import ASRModel
    from sonicscribe_amd.sessions import GatedSessions
    S = 128
    full = dims_name == "full"
    m = ASRModel.from_synthetic(spec.FULL if full else spec.TINY, device="cuda:0", max_batch=32, max_ctx=512)
    g = GatedSessions(m, [f"client-{i}" for i in range(S)])
    rng = np.random.default_rng(5)
    lead = rng.integers(3, 25, size=S)                 # silent chunks before the utterance
    n_speech = rng.integers(18, 60, size=S)            # 1.2 .. 3.8 s of "speech"
    n_ticks = int((lead + n_speech).max()) + 45
    wire = []
    for s in range(S):
        x = np.zeros(n_ticks * CHUNK, np.int16)
        sp = synth.synth_pcm(100 + s, int(n_speech[s]) * CHUNK).astype(np.float64)
        sp = sp / max(1.0, np.abs(sp).max()) * 30000.0                                      # loud: mean |s| / 12000 stays above the 0.9 cap
        x[lead[s] * CHUNK:(lead[s] + n_speech[s]) * CHUNK] = np.rint(sp).astype(np.int16)
        wire.append(x)
    events = []
    for t in range(n_ticks):
        for s in range(S):
            g.add_audio_chunk(s, wire[s][t * CHUNK:(t + 1) * CHUNK].tobytes(), timestamp=1000.0 + 0.064 * (t + 1))
        events.extend(g.tick(energy_vad, now=1000.0 + 0.064 * (t + 1)))
    finals = [e for e in events if e["type"] == "final"]
    partials = [e for e in events if e["type"] == "partial"]
    starts = [e for e in events if e["type"] == "speech_start"]
    assert len(starts) == S and len(finals) == S and len(partials) >= S       # one utterance per session, at least one partial each
    by_sess = {e["session"]: e for e in finals}
    assert len(by_sess) == S
    checked = 0
    for e in (finals[::9] + partials[::41] if full else finals + partials[::7]):
        s = int(e["session"].split("-")[1])
        a, n = e["first_sample"], e["n_samples"]
        # a final's budget follows segment_duration = min(audio length, timestamp span of the segment) (connection_manager.py:186-192)
        max_new = 15 if e["type"] == "partial" else min(50 + int(e["seconds"] * 5), 200)
        if e["type"] == "final":
            assert 0 < e["seconds"] <= n / 16000
        want = m.transcribe(frontend.pcm_bytes_to_float(wire[s][a:a + n].tobytes()), 16000, max_new_tokens=max_new)
        assert e["future"].result(timeout=120) == want, (e["type"], s)
        checked += 1
    # the final covers the utterance from the window in which the gate first saw speech (the reference's segment starts at that
    # window's first chunk, vad_processor_manager.py:127-129: up to one 640 ms window of the onset can precede it) to its end
    for e in finals:
        s = int(e["session"].split("-")[1])
        assert e["first_sample"] <= (lead[s] + 10) * CHUNK and e["first_sample"] + e["n_samples"] >= (lead[s] + n_speech[s]) * CHUNK
    batches = m._dispatcher.replicas[0].batches
    assert batches < len(finals) + len(partials) + checked                     # requests of a tick shared device batches
    g.close(); m.close()


def test_queued_max_length_request_survives_appends():
    """A 30 s final that waits in the queue while 64 ms chunks keep arriving (8 s of them here) must still find its oldest samples:
    the ring is larger than the visible buffer by a margin (asr.AudioStream)."""
    from sonicscribe_amd.asr import ASRModel
    m = ASRModel.from_synthetic(spec.TINY, device="cuda:0", max_batch=4, max_ctx=1024, slots=1)      # (one slot: the request below must queue behind the blocker)
    st = m.open_stream("c", buffer_seconds=30.0)
    raw = synth.synth_pcm(77, 38 * 16000)
    for i in range(0, 30 * 16000, CHUNK):
        st.add_audio_chunk(raw[i:i + CHUNK].tobytes())
    last = st.next_chunk_id - 1
    first, n = st.chunk_range_samples(0, last)
    assert n >= 29 * 16000
    eng = m.models[0]
    hold = threading.Event()
    orig = eng.transcribe_batch

    def slow(*a, **k):                                  # the replica is busy: the request below waits in the dispatcher queue
        hold.wait(30)
        return orig(*a, **k)
    eng.transcribe_batch = slow
    blocker = m.submit(raw[None, :16000].astype(np.float32) / 32768.0, 16000, 4)
    fut = st.submit_chunks(0, last, max_new_tokens=6)
    for i in range(30 * 16000, 38 * 16000, CHUNK):      # 8 s more arrive meanwhile: a 30 s ring would have lost the head of the request
        st.add_audio_chunk(raw[i:i + CHUNK].tobytes())
    hold.set()
    blocker.result(timeout=120)
    eng.transcribe_batch = orig
    want = m.transcribe(frontend.pcm_bytes_to_float(raw[first:first + n].tobytes()), 16000, max_new_tokens=6)
    assert fut.result(timeout=120) == want
    st.close(); m.close()
