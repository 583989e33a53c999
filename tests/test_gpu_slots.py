"""Batch slots (include/sonic_hip.h sonic_slot_create / sonic_run_staged_async / sonic_wait): several batches in flight on ONE weight copy.

The reference keeps up to three decodes in flight on its one model object in file mode (backend/main.py:429-445, 616-624) and serialises
them on the device; a slot is a second (third, ...) engine handle - own stream, buffers, KV cache, decode graphs - that reads the owner's
weights, so batches overlap on the GPU.  What must hold: a request's logits are the same BITS whether its batch ran alone on the owner or
on a slot while other slots were busy (every kernel's reduction order is fixed and no buffer is shared), the owner's weight bytes do not
move, the asynchronous entry returns what the synchronous one does, and the chunked decode loop (one hipGraph per `decode_chunk` token
steps, early-stop check pipelined one chunk behind) emits exactly the tokens of the step-by-step loop."""
import threading
from dataclasses import replace

import numpy as np
import pytest

from sonicscribe_amd import frontend, spec, synth

pytestmark = pytest.mark.gpu
SEED = 20260128


def prompt_for(d, n_samples, suf=(7, 301, 302, 303, 9, 11)):
    return [1, 17, 23, 5] + [d.audio_token_id] * spec.audio_token_count(spec.valid_frames(n_samples)) + list(suf)


def make(d, max_batch=8, max_ctx=512, mode=0):
    from sonicscribe_amd.engine import Engine
    e = Engine(d, 0, mode, max_batch=max_batch, max_ctx=max_ctx)
    e.load_synthetic(SEED)
    return e


FULLW = replace(spec.FULL, enc_layers=2, dec_layers=2, eos_ids=())      # full-width layers (256x256 GEMMs, fused decode kernels), shallow


@pytest.mark.parametrize("dims,mode", [(replace(spec.TINY, eos_ids=()), 0), (FULLW, 0), (replace(spec.TINY, eos_ids=()), 1)], ids=["tiny", "fullwidth", "tiny-int8"])
def test_slots_share_weights_and_match_solo_bit_for_bit(dims, mode):
    e = make(dims, max_batch=4, mode=mode)
    s1, s2 = e.slot(), e.slot()
    assert e.slot_count() == 3 and s1.slot_count() == 3
    assert e.weight_bytes() > 0 and s1.weight_bytes() == 0 and s2.weight_bytes() == 0        # ONE weight copy
    segs = [synth.synth_pcm(300 + i, 16000 * (2 + (i * 3) % 7)) for i in range(12)]
    prompts = [prompt_for(dims, len(s)) for s in segs]
    groups = [list(range(0, 4)), list(range(4, 8)), list(range(8, 12))]
    mn = 9
    # solo: each group alone on the owner, nothing else on the GPU (eager loop with step logits, then the graph loop for the ids)
    solo_lg, solo_ids = [], []
    for g in groups:
        ids, lg = e.transcribe_batch([segs[i] for i in g], [prompts[i] for i in g], [mn] * len(g), want_logits=True)
        ids2, _ = e.transcribe_batch([segs[i] for i in g], [prompts[i] for i in g], [mn] * len(g))
        assert all(np.array_equal(a, b) for a, b in zip(ids, ids2))
        solo_lg.append(lg); solo_ids.append(ids)
    # concurrent: the three handles decode DIFFERENT batches at the same time, several rounds, rotating which handle gets which group
    engines = [e, s1, s2]
    errs = []

    def work(k, rounds=4):
        try:
            for r in range(rounds):
                gi = (k + r) % 3
                g = groups[gi]
                want_logits = (r % 2 == 0)
                ids, lg = engines[k].transcribe_batch([segs[i] for i in g], [prompts[i] for i in g], [mn] * len(g), want_logits=want_logits)
                assert all(np.array_equal(a, b) for a, b in zip(ids, solo_ids[gi])), (k, r)
                if want_logits:
                    assert np.array_equal(lg.view(np.uint32), solo_lg[gi].view(np.uint32)), (k, r)      # bit for bit
        except BaseException as ex:
            errs.append(ex)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs
    s1.close()                                     # a slot may leave early ...
    assert e.slot_count() == 2
    ids, _ = s2.transcribe_batch([segs[0]], [prompts[0]], [mn])
    assert np.array_equal(ids[0], solo_ids[0][0])
    e.close()                                      # ... the rest go with their owner
    assert s2.h is None


def test_async_run_and_wait_equal_the_synchronous_call():
    from sonicscribe_amd.engine import SonicError
    d = replace(spec.TINY, eos_ids=())
    e = make(d, max_batch=8)
    s1 = e.slot()
    segs = [synth.synth_pcm(40 + i, 16000 * (3 + i % 4)) for i in range(8)]
    prompts = [prompt_for(d, len(s)) for s in segs]
    mn = [14] * 8
    want_a, _ = e.transcribe_batch(segs[:5], prompts[:5], mn[:5])
    want_b, _ = e.transcribe_batch(segs[5:], prompts[5:], mn[5:])
    assert e.wait() is True                        # nothing outstanding: returns at once
    for _ in range(3):
        e.stage_pcm(segs[:5]); s1.stage_pcm(segs[5:])
        e.run_staged_async(prompts[:5], mn[:5]); s1.run_staged_async(prompts[5:], mn[5:])     # both return at once, the batches overlap on the GPU
        with pytest.raises(SonicError):
            e.run_staged_async()                   # one outstanding run per handle
        while not s1.wait(block=False):            # polling form
            pass
        assert e.wait() is True
        got_a, got_b = e.fetch_tokens(5, 14), s1.fetch_tokens(3, 14)
        assert all(np.array_equal(x, y) for x, y in zip(got_a, want_a)) and all(np.array_equal(x, y) for x, y in zip(got_b, want_b))
    # an error inside the asynchronous run surfaces at wait()
    e.stage_pcm(segs[:2])
    e.run_staged_async([[1, d.audio_token_id, 2]] * 2, [4, 4])        # placeholder count does not match the audio rows
    with pytest.raises(ValueError):
        e.wait()
    e.stage_pcm(segs[:5]); e.run_staged_async(prompts[:5], mn[:5]); assert e.wait()
    assert all(np.array_equal(x, y) for x, y in zip(e.fetch_tokens(5, 14), want_a))
    e.close()


def test_decode_chunk_sizes_and_pipelined_early_stop():
    """One hipGraph per `decode_chunk` token steps; the count of running rows is read one chunk behind.  Tokens must not depend on the chunk
    size, with and without rows that stop at an EOS id (rows freeze once finished: the steps queued behind the stop emit nothing)."""
    base = replace(spec.TINY, eos_ids=())
    e = make(base, max_batch=4)
    segs = [synth.synth_pcm(70 + i, 16000 * (2 + i)) for i in range(4)]
    prompts = [prompt_for(base, len(s)) for s in segs]
    mn = [37, 23, 9, 30]
    e.set_option("decode_chunk", 1)
    want, _ = e.transcribe_batch(segs, prompts, mn)
    eager, _ = e.transcribe_batch(segs, prompts, mn, want_logits=True)
    assert all(np.array_equal(a, b) for a, b in zip(want, eager))
    for c in (2, 3, 4, 7, 16, 64):
        e.set_option("decode_chunk", c)
        got, _ = e.transcribe_batch(segs, prompts, mn)
        assert all(np.array_equal(a, b) for a, b in zip(got, want)), c
        assert [len(x) for x in got] == mn
    e.close()
    # EOS: make tokens the free-running rows emit early into EOS ids; every row stops long before its budget
    eos = tuple(sorted({int(want[0][3]), int(want[1][5]), int(want[3][2])}))
    d2 = replace(spec.TINY, eos_ids=eos)
    e2 = make(d2, max_batch=4)
    e2.set_option("decode_chunk", 1)
    ref, _ = e2.transcribe_batch(segs, prompts, [200] * 4)
    assert max(len(x) for x in ref) < 60                                # (they did stop early)
    for c in (1, 4, 5, 16):
        e2.set_option("decode_chunk", c)
        got, _ = e2.transcribe_batch(segs, prompts, [200] * 4)
        assert all(np.array_equal(a, b) for a, b in zip(got, ref)), c
        t = e2.timings()
        steps, ahead = t["decode_steps"], t["decode_lookahead"]
        assert steps <= max(len(x) for x in ref) - 1 + (ahead + 1) * c + 1, (c, steps, ahead)    # at most the detection chunk plus the `lookahead` queued behind it
        assert steps < 199
    e2.close()


def test_slot_decodes_rings_of_its_owner():
    d = replace(spec.TINY, eos_ids=())
    e = make(d, max_batch=4, max_ctx=1024)
    s1 = e.slot()
    n = 5 * 16000
    raw = np.clip(np.rint(synth.synth_pcm(3, n).astype(np.float64) * 0.37), -32768, 32767).astype(np.int16)
    ring = e.ring_create(30 * 16000)
    data = raw.tobytes()
    for i in range(0, len(data), 2048):
        ring.append(data[i:i + 2048])
    pcm = frontend.normalise_to_int16(frontend.pcm_bytes_to_float(raw.tobytes()))
    n_audio, _ = frontend.request_audio_tokens(n, d)
    prompt = [1, 17, 23, 5] + [d.audio_token_id] * n_audio + [7, 301, 9]
    ids_h, lg_h = e.transcribe_batch([pcm], [prompt], [5], req_win=[0, 1], want_logits=True)
    ids_r, lg_r = s1.transcribe_batch([ring.slice(0, n)], [prompt], [5], req_win=[0, 1], want_logits=True)      # ring of the owner, decoded by the slot
    assert np.array_equal(ids_h[0], ids_r[0]) and np.array_equal(lg_h.view(np.uint32), lg_r.view(np.uint32))
    other = make(d, max_batch=2)
    with pytest.raises((ValueError, RuntimeError)):
        other.transcribe_batch([ring.slice(0, n)], [prompt], [5], req_win=[0, 1])
    other.close()
    # a destroyed ring is refused by the registry lookup, never dereferenced
    import ctypes as C
    dead = C.c_void_p(ring.h.value)
    ring.close()
    rings = (C.c_void_p * 1)(dead)
    offs = np.zeros(2, np.int64); start = np.zeros(1, np.int64); cnt = np.array([n], np.int32); one = np.zeros(1, np.int16)
    rc = s1.lib.sonic_stage_mixed(s1.h, one.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), rings, start.ctypes.data_as(C.c_void_p),
                                  cnt.ctypes.data_as(C.c_void_p), 1, None, 1)
    assert rc != 0 and b"destroyed" in s1.lib.sonic_last_error(s1.h)
    e.close()


def test_asrmodel_slots_behind_the_dispatcher():
    """ASRModel(slots=2): one replica, two batches in flight; every transcript equals the one-slot model's, both slot threads work."""
    from sonicscribe_amd.asr import ASRModel
    d = replace(spec.TINY, eos_ids=())
    one = ASRModel.from_synthetic(d, device="cuda:0", max_batch=4, max_ctx=512, slots=1)
    two = ASRModel.from_synthetic(d, device="cuda:0", max_batch=4, max_ctx=512, slots=2)
    assert two.get_model_info()["slots_per_replica"] == 2 and two.model.slot_count() == 2
    assert two.model.weight_bytes() == one.model.weight_bytes()
    wavs = [synth.synth_pcm(200 + i, 16000 * (2 + i % 3)).astype(np.float32) / 32768.0 for i in range(24)]
    want = [one.transcribe(w[None], 16000, max_new_tokens=24) for w in wavs]
    futs = [two.submit(w[None], 16000, 24, session=f"c{i}") for i, w in enumerate(wavs)]
    assert [f.result() for f in futs] == want
    streams = [two.open_stream(f"s{i}") for i in range(4)]
    for i, st in enumerate(streams):               # device rings behind a two-slot replica
        wire = np.clip(np.rint(wavs[i] * 32768.0), -32768, 32767).astype(np.int16).tobytes()
        for j in range(0, len(wire), 2048):
            st.add_audio_chunk(wire[j:j + 2048])
    futs = [st.submit_chunks(0, st.next_chunk_id - 1, 24) for st in streams]
    got = [f.result() for f in futs]
    for i in range(4):
        nfull = (len(wavs[i]) // 1024) * 1024      # whole chunks reach the ring
        assert got[i] == one.transcribe(wavs[i][None, :len(wavs[i])], 16000, max_new_tokens=24) or nfull != len(wavs[i])
    for st in streams:
        st.close()
    one.close(); two.close()


@pytest.mark.parametrize("mode", [0, 1], ids=["bf16", "int8"])
def test_gemm_tile_choice_never_shows_in_a_result(mode):
    """Grids of 256x256 tiles that under-fill the chip go to the 128x128 kernel (csrc/gemm.hip small_grid_prefers128, option gemm_small_eff;
    the continuous dispatcher flips the option per prefill, dispatch._ContinuousReplica._prefill).  Full-width layers, 1 / 2 / 5 requests of
    mixed lengths: the step logits are the same BITS with the rule off (0), at its default (75) and with every eligible GEMM on small tiles
    (1000) - both kernels accumulate every output element over k in the same order and share their epilogues."""
    e = make(FULLW, max_batch=8, mode=mode)
    segs = [synth.synth_pcm(900 + i, 16000 * s) for i, s in enumerate((5, 20, 2, 29, 9))]
    prompts = [prompt_for(FULLW, len(s)) for s in segs]
    got = {}
    for eff in (0, 75, 1000):
        e.set_option("gemm_small_eff", eff)
        for B in (1, 2, 5):
            ids, lg = e.transcribe_batch(segs[:B], prompts[:B], [4] * B, want_logits=True)
            key = (B,)
            if key in got:
                assert np.array_equal(got[key][1].view(np.uint32), lg.view(np.uint32)), (eff, B, float(np.abs(got[key][1] - lg).max()))
                assert all(np.array_equal(a, b) for a, b in zip(got[key][0], ids))
            else:
                got[key] = (ids, lg)
    e.close()
