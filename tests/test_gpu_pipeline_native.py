"""sonic_pipeline_* (csrc/pipeline.cpp): the bulk pipeline's host loop as native threads inside the library - stage / prefill / splice / step /
fetch with condition variables and blocking events, no Python in the loop.  What must hold is what holds for the Python
driver it replaces (sonicscribe_amd/pipeline.py ContinuousPipeline, tests/test_gpu_continuous.py): every request's tokens are those of its solo
run, whatever shares its loop; plus the C-side contracts - a bad request fails alone, tickets can be waited for in any order, destroy completes
what was submitted.  The reference's counterpart: three executor threads around one model object (backend/main.py:429-445, 616-624)."""
from dataclasses import replace

import numpy as np
import pytest

from sonicscribe_amd import spec, synth

pytestmark = pytest.mark.gpu
SEED = 20260128


def prompt_for(d, n_samples, tail=()):
    return [1, 17, 23, 5] + [d.audio_token_id] * spec.audio_token_count(spec.valid_frames(n_samples)) + [7, 301, 302, 303, 9, 11] + list(tail)


@pytest.mark.parametrize("dims,mode,n_dec", [(replace(spec.TINY, eos_ids=()), 0, 2), (replace(spec.FULL, enc_layers=2, dec_layers=2, eos_ids=()), 0, 1),
                                            (replace(spec.TINY, eos_ids=()), 1, 2)], ids=["tiny-2dec", "fullwidth", "tiny-int8"])
def test_different_batches_equal_solo_runs(dims, mode, n_dec):
    from sonicscribe_amd.engine import Engine
    from sonicscribe_amd.pipeline import NativePipeline
    eng = Engine(dims, 0, mode, max_batch=8, max_ctx=512)
    eng.load_synthetic(SEED)
    slots = [eng.slot() for _ in range(n_dec + 1)]                       # n_dec decoding handles (the engine + slots) and two prefill slots
    decoders, prefills = [eng] + slots[:n_dec - 1], slots[n_dec - 1:]
    rng = np.random.default_rng(5)
    batches = []
    for b in range(7):
        R = 4 if b != 3 else 2                                           # one batch smaller than a block
        segs = [synth.synth_pcm(300 + 10 * b + i, 16000 * int(rng.integers(2, 12))) for i in range(R)]
        prompts = [prompt_for(dims, len(s), tail=[40 + b, 50 + i] * (1 + (b + i) % 3)) for i, s in enumerate(segs)]
        budgets = [int(rng.integers(1, 40)) for _ in range(R)]
        batches.append((segs, prompts, budgets))
    solo = [[prefills[-1].transcribe_batch([s], [p], [m])[0][0] for s, p, m in zip(*b)] for b in batches]
    pipe = NativePipeline(decoders, prefills, block=4)
    assert pipe.batches_in_flight == n_dec * 2 + len(prefills)
    tickets = [pipe.submit(p, m, segments=s) for s, p, m in batches]
    for b in (4, 0, 6, 1, 2, 3, 5):                                      # waited for in another order than submitted
        got = pipe.wait(tickets[b])
        assert len(got) == len(batches[b][1])
        for i, ids in enumerate(got):
            assert len(ids) == batches[b][2][i] and np.array_equal(ids, solo[b][i]), (b, i)
    # the benchmark's form: the batch every prefill handle has staged, submitted n times
    segs, prompts, budgets = batches[1]
    for p in prefills:
        p.stage_pcm(segs)
    res = pipe.run(9, prompts, budgets, lambda i, ids: np.array_equal(ids, solo[1][i]))
    assert res["batches"] == 9 and res["wrong_rows"] == 0 and res["decode_chunks"] > 0
    assert pipe.stats()["batches"] == 16
    pipe.close()
    again, _ = eng.transcribe_batch(batches[0][0], batches[0][1], batches[0][2])          # the decoders are batch engines again
    assert all(np.array_equal(a, s) for a, s in zip(again, solo[0]))
    eng.close()


def test_a_bad_request_fails_alone_and_destroy_completes_the_rest():
    from sonicscribe_amd.engine import Engine
    from sonicscribe_amd.pipeline import NativePipeline
    dims = replace(spec.TINY, eos_ids=())
    eng = Engine(dims, 0, 0, max_batch=4, max_ctx=512)
    eng.load_synthetic(SEED)
    pre = eng.slot()
    segs = [synth.synth_pcm(800 + i, 16000 * (3 + i)) for i in range(4)]
    prompts = [prompt_for(dims, len(s)) for s in segs]
    solo = [pre.transcribe_batch([s], [p], [12])[0][0] for s, p in zip(segs, prompts)]
    pipe = NativePipeline([eng], [pre], block=4)
    t_ok = pipe.submit(prompts, [12] * 4, segments=segs)
    bad_prompts = [p[:-3] if i != 2 else [q for q in p if q != dims.audio_token_id] for i, p in enumerate(prompts)]     # row 2: no audio placeholders
    t_bad = pipe.submit(bad_prompts, [12] * 4, segments=segs)
    t_ok2 = pipe.submit(prompts, [12] * 4, segments=segs)
    with pytest.raises(RuntimeError, match="audio tokens do not match"):
        pipe.wait(t_bad)
    for t in (t_ok, t_ok2):
        assert all(np.array_equal(a, s) for a, s in zip(pipe.wait(t), solo))
    with pytest.raises(RuntimeError):
        pipe.submit(prompts + prompts[:1], [12] * 5)                     # more requests than a block holds: refused at the boundary, nothing queued
    t_last = pipe.submit(prompts, [7] * 4, segments=segs)                # left un-waited: close() completes it
    keep = pipe._keep[t_last]
    pipe.close()
    assert all(np.array_equal(keep[2][r, :7], solo[r][:7]) and keep[3][r] == 7 for r in range(4))
    eng.close()


def test_create_validates_handles_and_a_device_error_fails_the_batch_loudly():
    """sonic_pipeline_create checks rows_per_decoder against every decoding handle's capacity, refuses a handle passed twice and handles of
    different weight copies (sonic_engine_info); a decode kernel that gives up on an in-kernel wait raises the device error word, which turns the
    step's running-row count negative - the batch (and, in a continuous loop, the pipeline) fails with a message instead of returning garbage."""
    import ctypes as C
    from sonicscribe_amd.engine import Engine
    from sonicscribe_amd.pipeline import NativePipeline
    dims = replace(spec.TINY, eos_ids=())
    eng = Engine(dims, 0, 0, max_batch=4, max_ctx=512)
    eng.load_synthetic(SEED)
    pre = eng.slot()
    other = Engine(dims, 0, 0, max_batch=4, max_ctx=512)
    other.load_synthetic(SEED)
    info = eng.info()
    assert info["max_batch"] == 4 and info["max_ctx"] == 512 and info["mode"] == 0 and info["weights_id"] == pre.info()["weights_id"] != other.info()["weights_id"]
    lib = eng.lib

    def create(dec, prefills, block, rows):
        d = (C.c_void_p * len(dec))(*[x.h for x in dec]); p = (C.c_void_p * len(prefills))(*[x.h for x in prefills]); h = C.c_void_p()
        rc = lib.sonic_pipeline_create(d, len(dec), p, len(prefills), block, rows, C.byref(h))
        if rc == 0:
            lib.sonic_pipeline_destroy(h)
        return rc
    assert create([eng], [pre], 4, 4) == 0
    assert create([eng], [pre], 4, 8) != 0            # rows_per_decoder above the handle's max_batch
    assert create([eng], [eng], 4, 4) != 0            # one handle as decoder and prefill slot
    assert create([eng], [other], 4, 4) != 0          # another weight copy
    # device error word
    seg = synth.synth_pcm(900, 16000 * 3)
    prompt = prompt_for(dims, len(seg))
    good = pre.transcribe_batch([seg], [prompt], [6])[0][0]
    pre.set_option("inject_dev_err", 1)
    with pytest.raises(RuntimeError, match="in-kernel wait"):
        pre.transcribe_batch([seg], [prompt], [6])
    pre.set_option("inject_dev_err", 0)
    assert np.array_equal(pre.transcribe_batch([seg], [prompt], [6])[0][0], good)
    pipe = NativePipeline([eng], [pre], block=4)
    assert np.array_equal(pipe.wait(pipe.submit([prompt], [6], segments=[seg]))[0], good)
    eng.set_option("inject_dev_err", 1)
    t = pipe.submit([prompt], [30], segments=[seg])
    with pytest.raises(RuntimeError, match="in-kernel wait"):
        pipe.wait(t)
    with pytest.raises(RuntimeError):
        pipe.submit([prompt], [6], segments=[seg])   # a failed pipeline refuses new work; its prefill threads sleep until close (no spin)
    pipe.close()
    eng.set_option("inject_dev_err", 0)
    other.close(); eng.close()
