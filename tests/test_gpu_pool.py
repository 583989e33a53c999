"""Device memory life cycle: uncached blocks are pooled while another engine lives on the device and handed back with the last one
(`del asr_model.model` + torch.cuda.empty_cache(), backend/main.py:84-90).  Own module: no module-scoped engine may be alive here."""
import numpy as np
import pytest

from sonicscribe_amd import spec, synth

pytestmark = pytest.mark.gpu


def tiny_prompt(n_samples, d=spec.TINY, pre=(1, 17, 23, 5), suf=(7, 301, 302, 303, 9, 11)):
    return list(pre) + [d.audio_token_id] * spec.audio_token_count(spec.valid_frames(n_samples)) + list(suf)


def test_pool_is_released_with_the_last_engine():
    """`del asr_model.model` (main.py:84-86) must give the memory back: the uncached blocks an engine parks in the process-wide pool are
    freed when the LAST engine of the device is destroyed, and kept for reuse while another one lives."""
    import gc
    from sonicscribe_amd.engine import Engine, device_info, release_pool
    gc.collect()
    a = Engine(spec.TINY, 0, max_batch=8, max_ctx=2048)
    a.load_synthetic(1)
    b = Engine(spec.TINY, 0, max_batch=2, max_ctx=512)
    b.load_synthetic(2)
    alloc_a, res_a = a.memory_info()
    a.close()                                           # b is alive: a's uncached blocks (KV cache, tiled weights ...) stay pooled
    alloc_b, res_b = b.memory_info()
    assert res_b - alloc_b > 0
    c = Engine(spec.TINY, 0, max_batch=8, max_ctx=2048)  # same sizes as a: takes them out of the pool again
    c.load_synthetic(1)
    assert b.memory_info()[1] - alloc_b < res_b - alloc_b
    segs = [synth.synth_pcm(3, 48000)]
    ids_c, _ = c.transcribe_batch(segs, [tiny_prompt(48000)], [6])
    ids_b, _ = b.transcribe_batch(segs, [tiny_prompt(48000)], [6])
    assert len(ids_c[0]) == 6 and len(ids_b[0]) == 6
    free_before = device_info(0)["free_bytes"]
    c.close(); b.close()
    assert release_pool(0) == 0                         # nothing left: the last destroy released the pool itself
    assert device_info(0)["free_bytes"] >= free_before + alloc_a // 2      # ... and the driver has the memory back
    d2 = Engine(spec.TINY, 0, max_batch=4, max_ctx=1024)  # a reload with other sizes works on recycled memory
    d2.load_synthetic(1)
    ids_d, _ = d2.transcribe_batch(segs, [tiny_prompt(48000)], [6])
    assert np.array_equal(ids_d[0], ids_c[0])
    d2.close()
