"""Device memory life cycle (`del asr_model.model`, backend/main.py:84-90): destroying an engine gives EVERYTHING back to the driver (round 4
retired the uncached activation buffers and their process-wide pool: measured worth nothing); slots return their buffers with themselves or
with their owner; a reload with other sizes works."""
import numpy as np
import pytest

from sonicscribe_amd import spec, synth

pytestmark = pytest.mark.gpu


def tiny_prompt(n_samples, d=spec.TINY, pre=(1, 17, 23, 5), suf=(7, 301, 302, 303, 9, 11)):
    return list(pre) + [d.audio_token_id] * spec.audio_token_count(spec.valid_frames(n_samples)) + list(suf)


def test_destroy_returns_weights_and_kv_cache():
    import gc
    from dataclasses import replace
    from sonicscribe_amd.engine import Engine, device_info
    gc.collect()
    d = replace(spec.FULL, enc_layers=2, dec_layers=4)             # full-width layers: hundreds of MB of weights, a 1 GiB KV cache
    free0 = device_info(0)["free_bytes"]
    a = Engine(d, 0, max_batch=32, max_ctx=2048)
    a.load_synthetic(1)
    alloc_a, res_a = a.memory_info()
    assert alloc_a > 2 ** 30 and device_info(0)["free_bytes"] < free0 - alloc_a // 2
    seg = [synth.synth_pcm(3, 48000)]
    n_audio = spec.audio_token_count(spec.valid_frames(48000))
    prompt = [1, 17, 23, 5] + [d.audio_token_id] * n_audio + [7, 301, 9]
    ids_a, _ = a.transcribe_batch(seg, [prompt], [6])
    a.close()
    free1 = device_info(0)["free_bytes"]
    assert free1 > free0 - 2 ** 28, (free0, free1)                  # everything but the runtime's own state (code objects, queues: < 256 MiB) is back
    b = Engine(d, 0, max_batch=8, max_ctx=512)                      # a reload with other sizes
    b.load_synthetic(1)
    alloc_b, res_b = b.memory_info()
    assert res_b == alloc_b                                         # no caching layer: reserved == allocated
    ids_b, _ = b.transcribe_batch(seg, [prompt], [6])
    assert np.array_equal(ids_a[0], ids_b[0])
    # slots: their buffers come and go with them, the weights stay one copy
    s1, s2 = b.slot(), b.slot()
    alloc_s, _ = s1.memory_info()
    assert 0 < alloc_s < alloc_b and b.memory_info()[0] == alloc_b and s1.weight_bytes() == 0
    free_with_slots = device_info(0)["free_bytes"]
    s1.close()
    assert device_info(0)["free_bytes"] > free_with_slots + alloc_s // 2
    b.close()                                                       # takes s2 with it
    assert s2.h is None
    assert device_info(0)["free_bytes"] > free0 - 2 ** 28
