"""CPU checks of the oracle's LLM.int8 restatement (oracle/sonic_oracle.c linear_int8, MODE_INT8).

PARITY UNPINNED: bitsandbytes (the third-party library backend/asr.py:182-198 swaps the linears for) is absent offline and is
CUDA-only, so nothing here is checked against the real library.  What IS pinned: the C restatement against an independent numpy
statement of the published algorithm (bitsandbytes 0.45-0.48 MatMul8bitLt.forward / int8_vectorwise_quant / int8_mixed_scaled_mm /
kdequant_mm_int32_fp16), and its structural properties.
"""
import numpy as np
import pytest

from oracle import oracle
from sonicscribe_amd import spec, synth


def f16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def numpy_linear8bit(x, w, bias, threshold=6.0):
    """Independent statement: row-wise int8 weights; activations per-row absmax without the outliers, outlier COLUMNS (any row >= 6)
    zeroed; int32 product; fp16(fma(acc, SCA*SCB/127^2, bias)); outlier columns added from x and the dequantised weights."""
    scb = np.abs(w).max(1).astype(np.float32)
    cb = np.rint(w * (np.float32(127) / scb)[:, None]).astype(np.int32)
    big = np.abs(x) >= threshold
    oc = big.any(0)
    sca = np.where(big, 0, np.abs(x)).max(1).astype(np.float32)
    ca = np.rint(x * (np.float32(127) / sca)[:, None]).astype(np.int32)
    ca[big] = 0; ca[:, oc] = 0
    c32 = (ca @ cb.T).astype(np.float64)
    sc = ((sca[:, None] * scb[None, :]).astype(np.float32) * np.float32(6.200012e-05)).astype(np.float64)
    y = f16((c32 * sc + (0 if bias is None else bias[None, :].astype(np.float64))).astype(np.float32))    # fma: one rounding to fp32
    if oc.any():
        wdq = f16(((cb.astype(np.float32) * scb[:, None]).astype(np.float32) * np.float32(7.874015718698502e-3)).astype(np.float32))
        a2 = np.zeros(y.shape, np.float32)
        for k in np.where(oc)[0]:
            a2 = (a2 + (x[:, k:k + 1] * wdq[None, :, k]).astype(np.float32)).astype(np.float32)
        y = f16((y + a2).astype(np.float32))
    return y


@pytest.mark.parametrize("T,N,K,outliers", [(37, 48, 256, 0), (64, 128, 384, 3), (1, 96, 512, 1), (130, 64, 128, 5)])
def test_linear_int8_equals_numpy_statement(T, N, K, outliers):
    rng = np.random.default_rng(T + N + K)
    x = f16(rng.standard_normal((T, K)) * 1.5); w = f16(rng.standard_normal((N, K)) * 0.05); b = f16(rng.standard_normal(N) * 0.1)
    for i in range(outliers):
        x[(7 * i) % T, (31 * i + 5) % K] = f16(6.0 + 1.7 * i) * (-1) ** i
    cb, scb = oracle.quantize_rows(w)
    y = oracle.linear_int8(x, cb, scb, b)
    ref = numpy_linear8bit(x, w, b)
    assert np.array_equal(y, ref), np.abs(y - ref).max()
    assert np.abs(y - (x @ w.T + b)).max() < 0.2          # and it approximates the dense product


def test_outlier_column_is_per_call():
    """A value >= 6 in one row removes its column from the int8 product for EVERY row of the call, not for other calls."""
    rng = np.random.default_rng(1)
    x = f16(rng.standard_normal((8, 128))); w = f16(rng.standard_normal((32, 128)) * 0.05)
    cb, scb = oracle.quantize_rows(w)
    base = oracle.linear_int8(x, cb, scb)
    x2 = x.copy(); x2[3, 40] = 9.0
    y2 = oracle.linear_int8(x2, cb, scb)
    # rows 0..2 of the call WITH the outlier row equal the numpy statement of the whole 8-row call (column 40 out of the int8 product for
    # every row, added back in fp16), and differ from the call without the outlier although only row 3 holds it
    assert np.array_equal(y2, numpy_linear8bit(x2, w, None)) and not np.array_equal(base[0], y2[0])
    without_col = oracle.linear_int8(np.delete(x2, 40, axis=1), *oracle.quantize_rows(np.delete(w, 40, axis=1)))
    assert np.abs(y2[:3] - without_col[:3] - f16(x2[:3, 40:41] * f16(cb[:, 40].astype(np.float32) * scb / 127.0)[None, :])).max() < 2e-2   # = int8 part without column 40 + its fp16 product
    alone = oracle.linear_int8(x2[:3], cb, scb)                                                # rows 0..2 as their own call: no outlier column
    assert np.array_equal(alone, base[:3])


def test_int8_model_modes():
    """MODE_INT8 = fp16 activations + Linear8bitLt on every swapped module (asr.py:169-210); lm_head / embedding / conv stem stay fp16."""
    d = spec.TINY
    st = synth.synth_state_dict(d, 20260128, 2)
    assert all(np.array_equal(v, f16(v)) for v in st.values())                                 # fp16(bf16(.)) weights: a bf16 checkpoint loaded as fp16
    m8 = oracle.Model(d, st, mode=oracle.MODE_INT8); m16 = oracle.Model(d, st, mode=oracle.MODE_FP16)
    pcm = synth.synth_pcm(10, 80000)
    feats, mask = oracle.logmel(pcm)
    prompt = [1, 17, 23, 5] + [d.audio_token_id] * spec.audio_token_count(int(mask.sum())) + [7, 301, 302, 303, 9, 11]
    r8 = m8.transcribe(feats, int(mask.sum()), prompt, 4, want=("enc_out",)); r16 = m16.transcribe(feats, int(mask.sum()), prompt, 4, want=("enc_out",))
    assert np.array_equal(r8["step_logits"], f16(r8["step_logits"]))                          # logits are fp16 values
    dl = np.abs(r8["step_logits"] - r16["step_logits"]).max()
    assert 1e-4 < dl < 0.3                                                                     # quantisation noise: visible, bounded
    e = np.abs(r8["enc_out"] - r16["enc_out"]); assert e.mean() < 0.05
