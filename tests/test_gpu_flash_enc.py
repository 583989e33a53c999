"""Round 5's encoder attention (csrc/attn_enc.hip: head dim 64, no mask - modeling_glmasr.py:171-221 through sdpa, driven from
backend/asr.py:411-422) in its three forms against the oracle's attention: the shipped one (option flash_enc = 1: running maximum fixed after
the first key tile), the classic online softmax of the same kernel (3), and the software-pipelined asm forms (5: one wave per SIMD, O and Q in
AGPRs; 6: two waves per SIMD, everything in VGPRs).  The fixed-maximum forms have a rare path - a partial row sum that leaves the safe range sends the tile
(form 1) or the block (form 5) through the exact computation - which a test has to FORCE (cdna_hip_programming.md rule 26): one late key
hundreds of nats above everything before it."""
import numpy as np
import pytest

from sonicscribe_amd import spec, synth

pytestmark = pytest.mark.gpu


def bf(x):
    return synth.round_bf16(np.asarray(x, np.float32))


@pytest.fixture(scope="module")
def eng():
    from sonicscribe_amd.engine import Engine
    e = Engine(spec.TINY, 0, max_batch=4, max_ctx=256)
    e.load_synthetic(20260128)
    yield e
    e.set_option("flash_enc", 1)
    e.close()


@pytest.fixture(scope="module")
def eng16():
    from sonicscribe_amd.engine import Engine, MODE_F16
    e = Engine(spec.TINY, 0, MODE_F16, max_batch=4, max_ctx=256)
    e.load_synthetic(20260128)
    yield e
    e.close()


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def ref_attention(orc, q, k, v):
    B, Tq, H, hd = q.shape
    Tk = k.shape[1]
    ref = np.empty_like(q)
    for b in range(B):
        o = np.empty((Tq, H, hd), np.float32)
        orc.lib().oracle_attention(q[b].ctypes.data, k[b].ctypes.data, v[b].ctypes.data, o.ctypes.data, Tq, Tk, H, H, hd, 0, Tk - Tq, 1)
        ref[b] = o
    return ref


FORMS = [1, 3, 5, 6]


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("Tq,Tk", [(1500, 1500), (100, 100), (64, 64), (65, 65), (257, 257), (300, 129), (40, 300), (256, 1), (1, 700)])
def test_shapes_against_oracle(eng, orc, form, Tq, Tk):
    """full and ragged key tiles (the last tile is masked), query counts that leave waves or whole lane halves without a row, Tq != Tk (per-sequence
    lengths through q_len / kv_len)"""
    rng = np.random.default_rng(Tq * 7 + Tk)
    B, H, hd = 2, 3, 64
    q = bf(rng.standard_normal((B, Tq, H, hd))); k = bf(rng.standard_normal((B, Tk, H, hd))); v = bf(rng.standard_normal((B, Tk, H, hd)))
    eng.set_option("flash_enc", form)
    got = eng.test_attention(q, k, v, False)
    ref = ref_attention(orc, q, k, v)
    err = np.abs(got - ref)
    assert np.isfinite(got).all() and err.max() <= 0.03 and err.mean() < 2e-3, (form, float(err.max()), float(err.mean()))


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("scale", [40.0, 400.0])
def test_late_dominant_key(eng, orc, form, scale):
    """scale 40: the late key sits ~40 / c above the first tile's maximum - huge probabilities, still inside the safe range (no rescue);
    scale 400: exp2 overflows to infinity - the rescue path must produce the exact row (and leave every other row of its block right)."""
    rng = np.random.default_rng(3)
    B, T, H, hd = 1, 700, 2, 64
    q = bf(rng.standard_normal((B, T, H, hd)) * 0.3); k = bf(rng.standard_normal((B, T, H, hd)) * 0.3); v = bf(rng.standard_normal((B, T, H, hd)))
    k[0, 517, 1] = bf(q[0, 10, 1] * scale)
    k[0, 69, 0] = bf(q[0, 300, 0] * scale)                   # and one in the second tile of the other head, seen from the second query block
    eng.set_option("flash_enc", form)
    got = eng.test_attention(q, k, v, False)
    ref = ref_attention(orc, q, k, v)
    err = np.abs(got - ref)
    assert np.isfinite(got).all() and err.max() <= 0.03, (form, scale, float(err.max()))
    # the dominated rows are (almost) a copy of one value row
    assert np.abs(got[0, 10, 1] - v[0, 517, 1]).max() <= 0.02 and np.abs(got[0, 300, 0] - v[0, 69, 0]).max() <= 0.02


def test_forms_agree_and_the_exact_form_matches_rounds_1_to_4(eng):
    """flash_enc 3 (exact maximum every tile) computes what flash_attn_kernel computes - same statistics, same rounding points - and must agree with it
    to the last bits almost everywhere; the fixed-maximum forms differ from it only by where each probability falls inside its bf16 binade."""
    rng = np.random.default_rng(11)
    B, T, H, hd = 2, 1500, 2, 64
    q = bf(rng.standard_normal((B, T, H, hd))); k = bf(rng.standard_normal((B, T, H, hd))); v = bf(rng.standard_normal((B, T, H, hd)))
    outs = {}
    for form in (0, 1, 3, 5, 6):
        eng.set_option("flash_enc", form)
        outs[form] = eng.test_attention(q, k, v, False)
    assert np.mean(outs[3] != outs[0]) < 1e-3 and np.abs(outs[3] - outs[0]).max() <= 2.0 ** -6
    for form in (1, 5, 6):
        d = np.abs(outs[form] - outs[0])
        assert d.max() <= 2.0 ** -5 and d.mean() < 1e-3, (form, float(d.max()), float(d.mean()))
    eng.set_option("flash_enc", 1)
    assert np.array_equal(outs[1], eng.test_attention(q, k, v, False))          # deterministic


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("scale", [8.0, 40.0])
def test_fp16_probabilities_do_not_overflow(eng16, form, scale):
    """The fp16 instantiation (int8 mode runs it: asr.py:61,296): a probability relative to the FIRST tile's maximum overflows fp16 at 65504, i.e.
    when a later score is ~11 nats above it - far inside what fp32 / bf16 tolerate.  The safe range of the fixed-maximum path is therefore per
    element type; found by tests/test_gpu_int8.py (NaN logits with LayerNorm weights x 3.5) before this test existed."""
    rng = np.random.default_rng(4)
    B, T, H, hd = 1, 400, 2, 64
    q = rng.standard_normal((B, T, H, hd)).astype(np.float32) * 0.3; k = rng.standard_normal((B, T, H, hd)).astype(np.float32) * 0.3
    v = rng.standard_normal((B, T, H, hd)).astype(np.float32)
    q, k, v = (x.astype(np.float16).astype(np.float32) for x in (q, k, v))
    k[0, 333, 1] = (q[0, 10, 1] * scale).astype(np.float16).astype(np.float32)
    eng16.set_option("flash_enc", form)
    got = eng16.test_attention(q, k, v, False)
    eng16.set_option("flash_enc", 1)
    s = np.einsum("bqhd,bkhd->bhqk", q.astype(np.float64), k.astype(np.float64)) / 8.0
    p = np.exp(s - s.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
    ref = np.einsum("bhqk,bkhd->bqhd", p, v.astype(np.float64))
    assert np.isfinite(got).all() and np.abs(got - ref).max() <= 0.01, (form, scale, float(np.abs(got - ref).max()))
