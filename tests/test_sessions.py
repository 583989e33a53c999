"""sessions.GatedSessions on CPU with stub streams: which decodes a tick triggers, over which chunk ranges, with which token budgets
(connection_manager.py:43-106,127-245, audio_manager.py:106-123, transcription_manager.py:19-41)."""
from concurrent.futures import Future

import numpy as np

from sonicscribe_amd.sessions import CHUNK_BYTES, GatedSessions, committed_max_new_tokens


class StubStream:
    def __init__(self):
        self.next_chunk_id, self.oldest_chunk_id, self.sub, self.ts = 0, 0, [], {}

    def add_audio_chunk(self, data, timestamp=None):
        self.ts[self.next_chunk_id] = timestamp
        self.next_chunk_id += 1
        return self.next_chunk_id - 1

    def chunk_timestamp(self, cid, default=0.0):
        return self.ts.get(cid, default)

    def chunk_range_samples(self, a, b):
        a = max(a, self.oldest_chunk_id)
        if b < a:
            raise ValueError("no audio")
        return a * 1024, (b - a + 1) * 1024

    def submit_samples(self, first, n, max_new, hotwords=None):
        self.sub.append((first, n, max_new))
        f = Future(); f.set_result(f"{first}:{n}:{max_new}")
        return f

    def close(self):
        pass


class StubModel:
    target_sr = 16000

    def open_stream(self, s, buffer_seconds):
        return StubStream()


def loud(v):
    return (np.full(1024, v, np.int16)).tobytes()


def energy_vad(rows, pcm, thr):
    return np.array([np.abs(p.astype(np.int32)).mean() / 10000.0 > t for p, t in zip(pcm, thr)])


def run(pattern, n_sessions=3, tick_every=1, vad=None):
    """pattern: per 64 ms, the sample value of the chunk every session receives (0 = silence); a tick after every `tick_every` chunks."""
    g = GatedSessions(StubModel(), [f"c{i}" for i in range(n_sessions)])
    ev = []
    for t, v in enumerate(pattern):
        for s in range(n_sessions):
            g.add_audio_chunk(s, loud(v), timestamp=100.0 + 0.064 * (t + 1))
        if (t + 1) % tick_every == 0:
            ev.append(g.tick(vad or energy_vad, now=100.0 + 0.064 * (t + 1)))
    return g, ev


def test_start_partials_and_final():
    # 10 silent chunks, 60 loud (3.84 s), 40 silent
    g, ev = run([0] * 10 + [8000] * 60 + [0] * 40)
    flat = [(t, e) for t, es in enumerate(ev) for e in es if e["session"] == "c0"]
    starts = [(t, e) for t, e in flat if e["type"] == "speech_start"]
    # the reference looks at the newest TWO chunks every tick, so after a window is consumed its last chunk re-enters the accumulator:
    # windows are chunks 0..9, 9..18, 18..27, ... (vad_processor_manager.py:64-66 + audio_manager.py:60-68); 9..18 is the first with speech
    assert len(starts) == 1 and starts[0][0] == 18 and starts[0][1]["start_chunk_id"] == 9
    partials = [(t, e) for t, e in flat if e["type"] == "partial"]
    assert partials[0][0] == 18                                                                    # first partial in the tick speech began
    assert all(b[0] - a[0] in (15, 16) for a, b in zip(partials, partials[1:]))                    # then at most once a second (15.6 ticks)
    for t, e in partials:
        assert e["start_chunk_id"] == max(9, t + 1 - 20) and e["end_chunk_id"] == t                # newest <= 20 chunks of the segment
        assert e["future"].result().endswith(":15")
    finals = [(t, e) for t, e in flat if e["type"] == "final"]
    # silence from chunk 70: windows 72..81 and 81..90 are silent -> speech ends in tick 90; the final covers chunk 9 .. newest (90)
    assert len(finals) == 1 and finals[0][0] == 90 and finals[0][1]["start_chunk_id"] == 9 and finals[0][1]["end_chunk_id"] == 90
    # budget: segment_duration = min(audio length, timestamp span chunk 9 -> chunk 90) = 81 * 64 ms (connection_manager.py:186-192), not the 82 chunks of audio
    assert abs(finals[0][1]["seconds"] - 81 * 0.064) < 1e-6
    assert finals[0][1]["n_samples"] == 82 * 1024 and finals[0][1]["future"].result() == f"{9 * 1024}:{82 * 1024}:{committed_max_new_tokens(81 * 0.064)}"
    assert committed_max_new_tokens(81 * 0.064) == 75 and committed_max_new_tokens(82 * 1024 / 16000) == 76     # (the two rules differ here)
    assert not [e for t, e in flat if e["type"] == "partial" and t > 90]                           # no partials once speech ended
    assert all(len([e for es in ev for e in es if e["session"] == f"c{i}"]) == len(flat) for i in range(3))   # every session alike


def test_long_segment_is_cut_at_30_s():
    n_loud = 520                                                                                    # 33.3 s of speech
    g, ev = run([0] * 10 + [9950] * n_loud + [0] * 30, n_sessions=1)       # louder than the threshold cap (0.9)
    finals = [e for es in ev for e in es if e["type"] == "final"]
    assert [e["parts"] for e in finals] == [2, 2]
    assert finals[0]["n_samples"] == 480000 and finals[0]["future"].result().endswith(":200")      # 30 s -> min(50 + 150, 200)
    rest = finals[1]["n_samples"]
    assert finals[1]["first_sample"] == finals[0]["first_sample"] + 480000 and rest == (finals[1]["end_chunk_id"] - 9 + 1) * 1024 - 480000
    assert 0 < rest < 480000
    assert finals[1]["future"].result().endswith(f":{committed_max_new_tokens(rest / 16000)}")


def test_span_of_30_s_or_less_is_one_request_even_when_the_audio_is_longer():
    """the split test uses segment_duration = min(actual, timestamp span), so audio of 30.02 s whose span is 29.95 s goes out as ONE request
    (the processor windows it), with the budget of the span (connection_manager.py:186-192)."""
    n_loud = 450                                            # speech ends at chunk 477: span 468 * 64 ms = 29.952 s, audio 469 chunks = 30.016 s
    g, ev = run([0] * 10 + [9950] * n_loud + [0] * 30, n_sessions=1)
    finals = [e for es in ev for e in es if e["type"] == "final"]
    assert len(finals) == 1 and finals[0]["parts"] == 1
    span = (finals[0]["end_chunk_id"] - 9) * 0.064
    assert finals[0]["n_samples"] / 16000 > 30.0 >= span
    assert abs(finals[0]["seconds"] - span) < 1e-6 and finals[0]["future"].result().endswith(f":{committed_max_new_tokens(span)}")


def test_slow_ticks_still_see_whole_windows():
    """a loop that ticks every 4 chunks: the accumulator holds ids older than the newest 14; the reference keeps the chunk OBJECTS in its
    accumulator (vad_processor_manager.py:64-66,85-86), so every window handed to the VAD has all 10 chunks of audio"""
    seen = []

    def vad(rows, pcm, thr):
        seen.extend(len(p) for p in pcm)
        return energy_vad(rows, pcm, thr)
    g, ev = run([0] * 10 + [8000] * 120 + [0] * 80, n_sessions=2, tick_every=4, vad=vad)
    assert seen and all(n == 10 * 1024 for n in seen)
    assert all(len(have) <= 16 for have in g.recent)        # and the byte store stays bounded
    assert [e["type"] for es in ev for e in es if e["session"] == "c0"].count("final") == 1


def test_chunk_size_constant():
    assert CHUNK_BYTES == int(16000 * 2 * 64 / 1000)           # config.py:24
