"""Device-resident ingest (include/sonic_hip.h sonic_ring_*): a decode whose windows are slices of a session's PCM ring
in HBM must give exactly what the host path gives on the same wire bytes -- the reference's conversions (int16 -> float / 32768,
transcription_manager.py:45-54; peak normalisation + PCM_16 round trip, asr.py:247-276) restated in numpy by
frontend.pcm_bytes_to_float / normalise_to_int16 on one side, done by csrc/ingest.hip on the other.  One differing sample would change
the log-mel features and with them the logits, so logits are compared bit for bit."""
import numpy as np
import pytest

from sonicscribe_amd import frontend, spec, synth

pytestmark = pytest.mark.gpu

CHUNK = 2048      # AUDIO_CHUNK_SIZE bytes (backend/config.py:24): 64 ms of 16 kHz int16


def host_windows(raw: np.ndarray, dims):
    """the host path of asr.py for one request: bytes -> float -> normalised int16 -> 30 s windows"""
    pcm = frontend.normalise_to_int16(frontend.pcm_bytes_to_float(raw.tobytes()))
    return [pcm[s:e] for s, e in frontend.split_windows(len(pcm), dims)]


def prompt_for(dims, n_samples):
    n_audio, _ = frontend.request_audio_tokens(n_samples, dims)
    return [1, 17, 23, 5] + [dims.audio_token_id] * n_audio + [7, 301, 9]


@pytest.fixture(scope="module")
def eng():
    from dataclasses import replace
    from sonicscribe_amd.engine import Engine
    e = Engine(replace(spec.TINY, eos_ids=()), 0, max_batch=4, max_ctx=1024)
    e.load_synthetic(11)
    yield e
    e.close()


def wire(seed, n, scale):
    """raw wire samples: NOT peak-normalised (the normalisation is what is under test)"""
    x = synth.synth_pcm(seed, n).astype(np.float64) * scale
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


@pytest.mark.parametrize("seconds,scale", [(5.0, 0.37), (1.28, 0.05), (20.0, 1.0)])
def test_ring_equals_host_path(eng, seconds, scale):
    n = int(seconds * 16000)
    raw = wire(3, n, scale)
    ring = eng.ring_create(30 * 16000)
    data = raw.tobytes()
    first = [ring.append(data[i:i + CHUNK]) for i in range(0, len(data), CHUNK)]
    assert first[0] == 0 and ring.head == n
    prompt = prompt_for(eng.dims, n)
    ids_h, lg_h = eng.transcribe_batch(host_windows(raw, eng.dims), [prompt], [5], req_win=[0, 1], want_logits=True)
    ids_r, lg_r = eng.transcribe_batch([ring.slice(0, n)], [prompt], [5], req_win=[0, 1], want_logits=True)
    assert np.array_equal(ids_h[0], ids_r[0])
    assert np.array_equal(lg_h.view(np.uint32), lg_r.view(np.uint32))
    ring.close()


def test_ring_equals_host_path_int8_mode():
    """the same in INT8 mode (fp16 hand-off of the log-mel features, LLM.int8 linears): staging is mode-independent, the results must
    again be the same bits"""
    from dataclasses import replace
    from sonicscribe_amd.engine import Engine, MODE_INT8
    e = Engine(replace(spec.TINY, eos_ids=()), 0, MODE_INT8, max_batch=4, max_ctx=1024)
    e.load_synthetic(11)
    n = 7 * 16000 + 321
    raw = wire(13, n, 0.21)
    ring = e.ring_create(30 * 16000)
    data = raw.tobytes()
    for i in range(0, len(data), CHUNK):
        ring.append(data[i:i + CHUNK])
    prompt = prompt_for(e.dims, n)
    ids_h, lg_h = e.transcribe_batch(host_windows(raw, e.dims), [prompt], [5], want_logits=True)
    ids_r, lg_r = e.transcribe_batch([ring.slice(0, n)], [prompt], [5], want_logits=True)
    assert np.array_equal(ids_h[0], ids_r[0]) and np.array_equal(lg_h.view(np.uint32), lg_r.view(np.uint32))
    e.close()


def test_ring_wraps_and_rejects_what_it_lost(eng):
    cap = 3 * 16000
    ring = eng.ring_create(cap)
    raw = wire(5, 5 * 16000, 0.6)
    for i in range(0, len(raw), 1000):                       # chunk size that does not divide the capacity: appends split at the wrap
        ring.append(raw[i:i + 1000])
    assert ring.head == len(raw)
    start, n = len(raw) - 2 * 16000 - 123, 2 * 16000 + 123     # the newest 2 s: spans the physical end of the ring
    assert start // cap != (start + n - 1) // cap
    prompt = prompt_for(eng.dims, n)
    ids_h, lg_h = eng.transcribe_batch(host_windows(raw[start:start + n], eng.dims), [prompt], [4], want_logits=True)
    ids_r, lg_r = eng.transcribe_batch([ring.slice(start, n)], [prompt], [4], want_logits=True)
    assert np.array_equal(ids_h[0], ids_r[0]) and np.array_equal(lg_h.view(np.uint32), lg_r.view(np.uint32))
    with pytest.raises(RuntimeError, match="not in the ring"):
        eng.transcribe_batch([ring.slice(0, 16000)], [prompt_for(eng.dims, 16000)], [2])           # overwritten long ago
    with pytest.raises(RuntimeError, match="not in the ring"):
        eng.transcribe_batch([ring.slice(len(raw) - 100, 200)], [prompt_for(eng.dims, 200)], [2])  # not appended yet
    ring.close()


def test_multi_window_request_shares_one_peak(eng):
    """a 47 s request = two 30 s windows; the peak sits in the second window, the first must be scaled by it too"""
    n = 47 * 16000
    raw = wire(8, n, 0.2)
    raw[40 * 16000 + 7] = -32768                              # the extreme sample (|s| = 32768 -> m = 1.0)
    ring = eng.ring_create(60 * 16000)
    ring.append(raw)
    wins = frontend.split_windows(n, eng.dims)
    assert len(wins) == 2
    prompt = prompt_for(eng.dims, n)
    ids_h, lg_h = eng.transcribe_batch(host_windows(raw, eng.dims), [prompt], [4], req_win=[0, 2], want_logits=True)
    ids_r, lg_r = eng.transcribe_batch([ring.slice(s, e - s) for s, e in wins], [prompt], [4], req_win=[0, 2], want_logits=True)
    assert np.array_equal(ids_h[0], ids_r[0]) and np.array_equal(lg_h.view(np.uint32), lg_r.view(np.uint32))
    ring.close()


def test_mixed_batch_and_silence(eng):
    """one batch with a ring request, a host request and an all-zero ring request (m <= 1e-6: passed through unnormalised)"""
    n1, n2, n3 = 3 * 16000, 4 * 16000 + 11, 2 * 16000
    raw1, raw2 = wire(21, n1, 0.5), wire(22, n2, 0.8)
    r1, r3 = eng.ring_create(10 * 16000), eng.ring_create(10 * 16000)
    r1.append(raw1); r3.append(np.zeros(n3, np.int16))
    prompts = [prompt_for(eng.dims, n1), prompt_for(eng.dims, n2), prompt_for(eng.dims, n3)]
    segs_mixed = [r1.slice(0, n1)] + host_windows(raw2, eng.dims) + [r3.slice(0, n3)]
    segs_host = host_windows(raw1, eng.dims) + host_windows(raw2, eng.dims) + [np.zeros(n3, np.int16)]
    ids_m, lg_m = eng.transcribe_batch(segs_mixed, prompts, [4, 4, 4], want_logits=True)
    ids_h, lg_h = eng.transcribe_batch(segs_host, prompts, [4, 4, 4], want_logits=True)
    assert all(np.array_equal(a, b) for a, b in zip(ids_m, ids_h))
    assert np.array_equal(lg_m.view(np.uint32), lg_h.view(np.uint32))
    other = None
    try:                                                      # a ring can only be decoded by its own engine
        from sonicscribe_amd.engine import Engine
        other = Engine(eng.dims, 0, max_batch=2, max_ctx=512)
        other.load_synthetic(11)
        with pytest.raises(ValueError, match="owns the ring"):
            other.transcribe_batch([r1.slice(0, n1)], [prompts[0]], [2])
    finally:
        if other is not None:
            other.close()
    r1.close(); r3.close()


def test_audio_stream_facade_equals_transcribe():
    """AudioStream = the reference's chunk store on the device: add_audio_chunk per 64 ms wire chunk, partial = the newest 20 chunks
    (audio_manager.py:99-114), final = the whole segment (:115-123); results equal transcribe() on the concatenated bytes."""
    import asyncio
    from sonicscribe_amd.asr import ASRModel
    m = ASRModel.from_synthetic(spec.TINY, device="cuda:0,0", max_batch=8, max_ctx=512)     # two replicas: the stream pins its requests
    raw = wire(31, 6 * 16000, 0.4)
    data = raw.tobytes()
    st = m.open_stream("client-42")
    ids = [st.add_audio_chunk(data[i:i + CHUNK]) for i in range(0, len(data), CHUNK)]
    assert ids == list(range(len(ids)))
    part = st.submit_chunks(ids[-20], ids[-1], max_new_tokens=15)
    fin = st.submit_chunks(ids[0], ids[-1], max_new_tokens=20, hotwords=["alpha"])
    want_part = m.transcribe(frontend.pcm_bytes_to_float(data[-20 * CHUNK:]), 16000, max_new_tokens=15)
    want_fin = m.transcribe(frontend.pcm_bytes_to_float(data), 16000, max_new_tokens=20, hotwords=["alpha"])
    assert part.result() == want_part and fin.result() == want_fin
    assert asyncio.run(st.transcribe_chunks(ids[-20], ids[-1], 15)) == want_part
    with pytest.raises(ValueError):
        st.submit_chunks(len(ids) + 5, len(ids) + 9)
    st.close()
    # a buffer shorter than the audio: chunks that left the ring are skipped, as get_chunks_by_range skips the ids that
    # _cleanup_old_chunks removed (audio_manager.py:35-59, 76-79)
    short = m.open_stream("client-43", buffer_seconds=2.0)
    ids = [short.add_audio_chunk(data[i:i + CHUNK]) for i in range(0, len(data), CHUNK)]
    kept = sorted(short._chunks)
    assert kept[-1] == ids[-1] and 0 < len(kept) <= 2 * 16000 // 1024 and kept == list(range(kept[0], kept[-1] + 1))
    got = short.submit_chunks(0, ids[-1], max_new_tokens=10).result()
    assert got == m.transcribe(frontend.pcm_bytes_to_float(data[kept[0] * CHUNK:]), 16000, max_new_tokens=10)
    short.close()
    m.close()


def test_small_ring_bursts(eng):
    """a ring barely larger than one window, fed in bursts that lap it several times between decodes: the pinned mirror must not be
    overwritten under a copy that has not left yet"""
    cap = 20000
    ring = eng.ring_create(cap)
    raw = wire(41, 16000 * 8, 0.7)
    for i in range(0, len(raw), 7000):                        # 7000-sample appends: the ring is lapped every three of them
        ring.append(raw[i:i + 7000])
    n = 18000
    start = len(raw) - n
    prompt = prompt_for(eng.dims, n)
    ids_h, lg_h = eng.transcribe_batch(host_windows(raw[start:], eng.dims), [prompt], [3], want_logits=True)
    ids_r, lg_r = eng.transcribe_batch([ring.slice(start, n)], [prompt], [3], want_logits=True)
    assert np.array_equal(ids_h[0], ids_r[0]) and np.array_equal(lg_h.view(np.uint32), lg_r.view(np.uint32))
    ring.close()


def test_rings_die_with_their_engine():
    """sonic_destroy frees the rings that are still alive (C-level contract in include/sonic_hip.h); the Python wrappers close the
    rings first, and closing a stream after its model is a no-op."""
    from sonicscribe_amd.asr import ASRModel
    from sonicscribe_amd.engine import Engine
    e = Engine(spec.TINY, 0, max_batch=2, max_ctx=512)
    e.load_synthetic(3)
    rings = [e.ring_create(16000) for _ in range(3)]
    for r in rings:
        r.append(np.arange(1000, dtype=np.int16))
    rings[0].close()                                   # one destroyed by the caller, two left to the engine
    e.lib.sonic_destroy(e.h)                           # the C call itself, not Engine.close()
    e.h = None
    for r in rings:
        r.h = None                                     # (their handles are dead now)
    m = ASRModel.from_synthetic(spec.TINY, max_batch=2, max_ctx=512)
    st = m.open_stream("c")
    st.add_audio_chunk(np.zeros(1024, np.int16).tobytes())
    m.close()
    st.close()                                         # after the model: nothing left to free, must not crash
