"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Tolerances (stated per test):
  * log-mel front-end (fp32): |diff| <= 1e-3 as north_star asks (observed ~1e-5).
  * bf16 kernels: the oracle reproduces torch's op-boundary rounding; what remains is fp32 accumulation order, i.e.
    isolated 1-ulp bf16 flips -> tolerance 2 bf16 ulp of the output magnitude for single ops, and a bf16-derived logit bound
    (stated in the test) end to end; token IDs must be bit-exact wherever the reference top-1/top-2 margin exceeds it.
"""
import os

import numpy as np
import pytest

from sonicscribe_amd import spec, synth

pytestmark = pytest.mark.gpu


def bf(x):
    return synth.round_bf16(np.asarray(x, np.float32))


@pytest.fixture(scope="module")
def eng():
    from sonicscribe_amd.engine import Engine
    e = Engine(spec.TINY, 0, max_batch=16, max_ctx=512)
    e.load_synthetic(20260128)
    yield e
    e.close()


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def ulp_tol(ref, n_ulp=2.0):
    return n_ulp * np.maximum(np.abs(ref), 1e-2) * 2.0 ** -8


def assert_bf16_close(got, ref, what, n_ulp_of_max=2.0, min_exact=0.4, max_mean=5e-3):
    """Activation tensors of a bf16 layer stack: the HIP path and the oracle round at the same op boundaries, so most elements agree
    BIT FOR BIT and the rest differ by isolated bf16 flips.  Bounds (measured with tools/measure_enc_err.py: 50-67 % of the elements
    exact, max |diff| = 1-2 ulp of the tensor's largest magnitude, mean |diff| 1.4e-3 .. 3.6e-3):
      * max |diff| <= n_ulp_of_max bf16 ulps of max |ref|;  * at least min_exact of the elements identical;  * mean |diff| <= max_mean."""
    d = np.abs(got - ref)
    top = float(np.abs(ref).max())
    ulp_top = 2.0 ** (np.floor(np.log2(max(top, 2.0 ** -8))) - 7)
    exact = float(np.mean(got == ref))
    assert d.max() <= n_ulp_of_max * ulp_top and exact >= min_exact and d.mean() <= max_mean, \
        (what, float(d.max()), n_ulp_of_max * ulp_top, exact, float(d.mean()))


# ------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 384, 128), (1500, 256, 384), (33, 132, 64)])
def test_gemm_bias(eng, orc, M, N, K):
    rng = np.random.default_rng(M + N + K)
    A = bf(rng.standard_normal((M, K)) * 0.5); W = bf(rng.standard_normal((N, K)) * 0.2); b = bf(rng.standard_normal(N) * 0.1)
    got = eng.test_gemm(A, W, b)
    ref = bf((A.astype(np.float64) @ W.T.astype(np.float64) + b).astype(np.float32))
    assert np.all(np.abs(got - ref) <= ulp_tol(ref)), np.abs(got - ref).max()


def test_gemm_identity_asymmetric(eng):
    # A = I with an asymmetric W catches a swapped row/col map (cdna_hip_programming.md §3)
    K = 128
    A = np.eye(K, dtype=np.float32); W = bf(np.arange(256 * K, dtype=np.float32).reshape(256, K) % 251 / 16.0)
    got = eng.test_gemm(A, W)
    assert np.array_equal(got, W.T)


def test_gemm_epilogues(eng, orc):
    from sonicscribe_amd.engine import EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_SWIGLU
    rng = np.random.default_rng(5)
    M, N, K = 300, 256, 128
    A = bf(rng.standard_normal((M, K))); W = bf(rng.standard_normal((N, K)) * 0.1); b = bf(rng.standard_normal(N) * 0.1)
    R = bf(rng.standard_normal((M, N)))
    lin = bf((A.astype(np.float64) @ W.T.astype(np.float64) + b).astype(np.float32))
    import math
    erf = np.vectorize(math.erf)
    gelu = bf((0.5 * lin * (1.0 + erf(lin.astype(np.float64) / math.sqrt(2.0)))).astype(np.float32))
    got = eng.test_gemm(A, W, b, epi=EPI_BIAS_GELU)
    assert np.all(np.abs(got - gelu) <= ulp_tol(gelu, 3)), np.abs(got - gelu).max()
    got = eng.test_gemm(A, W, b, resid=R, epi=EPI_BIAS_RESID)
    ref = bf(lin + R)
    assert np.all(np.abs(got - ref) <= ulp_tol(ref, 3))
    # SwiGLU: rows interleaved gate/up in groups of 16
    ff = N // 2
    Wg = bf(rng.standard_normal((ff, K)) * 0.1); Wu = bf(rng.standard_normal((ff, K)) * 0.1)
    Wi = np.empty((N, K), np.float32)
    for g in range(ff // 16):
        Wi[32 * g:32 * g + 16] = Wg[16 * g:16 * g + 16]; Wi[32 * g + 16:32 * g + 32] = Wu[16 * g:16 * g + 16]
    got = eng.test_gemm(A, Wi, epi=EPI_SWIGLU)
    gg = bf((A.astype(np.float64) @ Wg.T.astype(np.float64)).astype(np.float32)); uu = bf((A.astype(np.float64) @ Wu.T.astype(np.float64)).astype(np.float32))
    ref = bf(bf(gg / (1.0 + np.exp(-gg))) * uu)
    assert np.all(np.abs(got - ref) <= ulp_tol(ref, 4) + 1e-3), np.abs(got - ref).max()


@pytest.mark.parametrize("M,N,K,epi", [(1024, 512, 256, 0), (777, 384, 512, 1), (2048, 256, 1280, 2), (600, 1280, 320, 0), (520, 512, 256, 3),
                                       (8320, 2048, 256, 2), (8300, 2048, 256, 3)])   # last two: ragged tail rows cut off to the 128x128 kernel
def test_gemm256_path(eng, M, N, K, epi):
    """Shapes large enough for the 256x256 multi-phase kernel (M >= 512, N >= 256, K >= 256), ragged edges included."""
    import math
    rng = np.random.default_rng(M + N)
    A = bf(rng.standard_normal((M, K)) * 0.5); W = bf(rng.standard_normal((N, K)) * 0.2); b = bf(rng.standard_normal(N) * 0.1)
    lin = lambda bias: bf((A.astype(np.float64) @ W.T.astype(np.float64) + bias).astype(np.float32))
    if epi == 0:
        got, ref = eng.test_gemm(A, W, b), lin(b)
    elif epi == 1:
        l = lin(b); erf = np.vectorize(math.erf)
        got, ref = eng.test_gemm(A, W, b, epi=1), bf((0.5 * l * (1.0 + erf(l.astype(np.float64) / math.sqrt(2.0)))).astype(np.float32))
    elif epi == 2:
        R = bf(rng.standard_normal((M, N)))
        got, ref = eng.test_gemm(A, W, b, resid=R, epi=2), bf(lin(b) + R)
        ref_mag = np.abs(lin(b)) + np.abs(R)      # the residual add can cancel: tolerance follows the addends
    else:
        ff = N // 2
        g = lin(0.0)
        idx_g = np.concatenate([np.arange(32 * i, 32 * i + 16) for i in range(ff // 16)]); idx_u = idx_g + 16
        gg, uu = g[:, idx_g], g[:, idx_u]
        got, ref = eng.test_gemm(A, W, epi=3), bf(bf(gg / (1.0 + np.exp(-gg))) * uu)
    bad = np.abs(got - ref) > ulp_tol(ref_mag if epi == 2 else ref, 4) + 1e-3
    if bad.any():      # footprint of the failure: which tiles, and what the wrong values look like
        rows = np.where(bad.any(1))[0]; cols = np.where(bad.any(0))[0]
        again = eng.test_gemm(A, W, None if epi == 3 else b, resid=(R if epi == 2 else None), epi=epi)
        msg = (f"{int(bad.sum())} bad elements, max err {np.abs(got - ref).max():.3f}; rows {rows.min()}..{rows.max()} ({len(rows)}), cols {cols.min()}..{cols.max()} "
                             f"({len(cols)}); row tiles {sorted(set((rows // 256).tolist()))[:12]}, col tiles {sorted(set((cols // 256).tolist()))[:12]}; "
                             f"bad values are zero: {bool(np.all(got[bad] == 0))}; a rerun is clean: {bool(not (np.abs(again - ref) > ulp_tol(ref_mag if epi == 2 else ref, 4) + 1e-3).any())}; "
                             f"rerun identical to first: {bool(np.array_equal(again, got))}; first bad (row, col, got, ref): "
                             f"{[(int(r), int(c), float(got[r, c]), float(ref[r, c])) for r, c in list(zip(*np.where(bad)))[:6]]}")
        try:      # keep the footprint of this rare, box-dependent failure where the driver's log tail cannot lose it
            os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
            with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "gemm256_flake.txt"), "a") as fh:
                fh.write(f"M={M} N={N} K={K} epi={epi}: {msg}\n")
        except OSError:
            pass
        raise AssertionError(msg)
    eng.set_option("gemm_force128", 1)
    try:
        old = eng.test_gemm(A, W, None if epi == 3 else b, resid=(R if epi == 2 else None), epi=epi)
    finally:
        eng.set_option("gemm_force128", 0)
    # Both kernels accumulate every output element over k in the same order (32-wide MFMA steps, ascending) and share the epilogue
    # arithmetic (the GELU table included), so a row's result does not depend on which kernel its batch size selects: bit-identical.
    assert np.array_equal(old.view(np.uint32), got.view(np.uint32)), float(np.mean(old != got))


def test_gelu_table_bit_exact(eng, orc):
    """The fc1 epilogue looks GELU up in an LDS table indexed by the bf16 bits of the pre-activation.  With operands whose products and
    sums are exact in fp32 (small dyadic values) the pre-activation does not depend on the summation order, so the output must equal
    the oracle's GELU bit for bit -- over the table range, below it (|x| < 2^-14 -> 0.5 x) and above it (|x| >= 16 -> x or -0)."""
    rng = np.random.default_rng(5)
    M, N, K = 512, 512, 256
    A = rng.integers(-4, 5, (M, K)).astype(np.float32) / 4.0
    W = rng.integers(-2, 3, (N, K)).astype(np.float32) / 8.0
    b = np.zeros(N, np.float32)
    b[:64] = 2.0 ** -16; b[64:128] = -2.0 ** -15                 # columns whose bias alone lands below the table on zero sums
    W[:128, 8:] = 0.0                                            # (few terms: many exact zeros and tiny sums there)
    W[128:192] *= 8.0                                            # columns that leave the table at the top (|x| >= 16)
    l = bf((A.astype(np.float64) @ W.T.astype(np.float64) + b).astype(np.float32))
    assert (np.abs(l) >= 16).any() and ((np.abs(l) < 2.0 ** -14) & (l != 0)).any() and (l == 0).any()
    got = eng.test_gemm(A, W, b, epi=1)
    ref = orc.gelu(l, bf16=True)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (int((got != ref).sum()), np.abs(got - ref).max())
    eng.set_option("no_gelu_lut", 1)
    try:
        arith = eng.test_gemm(A, W, b, epi=1)
    finally:
        eng.set_option("no_gelu_lut", 0)
    # the arithmetic epilogue uses the erfc form for x < 0 (no cancellation), the reference op order cancels in fp32 there: the two
    # agree to an ulp except in that tail, where the values are < 2e-3 in magnitude
    d = np.abs(arith - got)
    assert np.all(d <= ulp_tol(got, 1) + 2e-4), d.max()


@pytest.mark.parametrize("M,N,K", [(1, 64, 256), (32, 256, 256), (32, 1024, 512), (40, 512, 2048), (64, 128, 768), (33, 48, 1024), (32, 2048, 6144),
                                   (32, 3072, 2048), (17, 2048, 6144), (64, 3072, 2048), (5, 4096, 1024)])   # one-block-per-CU tilings
def test_skinny(eng, M, N, K):
    rng = np.random.default_rng(M * 7 + N)
    X = bf(rng.standard_normal((M, K))); W = bf(rng.standard_normal((N, K)) * 0.1)
    got = eng.test_skinny(X, W)
    ref = (X.astype(np.float64) @ W.T.astype(np.float64)).astype(np.float32)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-3 * np.sqrt(K) * 0.1)


@pytest.mark.parametrize("M,ff,K", [(32, 6144, 2048), (5, 4096, 256), (17, 4096, 512), (64, 6144, 2048), (41, 4096, 512), (33, 4096, 1024), (50, 4096, 256)])
def test_skinny_fused_gate_up(eng, M, ff, K):      # 33 .. 64 rows: skinny_gu64_kernel (one sweep; K = 256: the two-pass form)
    rng = np.random.default_rng(ff + K + M)
    X = bf(rng.standard_normal((M, K))); Wg = bf(rng.standard_normal((ff, K)) * 0.05); Wu = bf(rng.standard_normal((ff, K)) * 0.05)
    Wi = np.empty((2 * ff, K), np.float32)
    for g in range(ff // 16):
        Wi[32 * g:32 * g + 16] = Wg[16 * g:16 * g + 16]; Wi[32 * g + 16:32 * g + 32] = Wu[16 * g:16 * g + 16]
    got = eng.test_skinny_gu(X, Wi)
    gg = bf((X.astype(np.float64) @ Wg.T.astype(np.float64)).astype(np.float32)); uu = bf((X.astype(np.float64) @ Wu.T.astype(np.float64)).astype(np.float32))
    ref = bf(bf(gg / (1.0 + np.exp(-gg))) * uu)
    assert np.all(np.abs(got - ref) <= ulp_tol(ref, 4) + 1e-3), np.abs(got - ref).max()


@pytest.mark.parametrize("hd,Hq,Hkv,Tq,Tk,causal", [(64, 2, 2, 1500, 1500, False), (64, 3, 3, 100, 100, False), (128, 4, 1, 264, 264, True),
                                                    (128, 2, 1, 72, 72, True), (128, 4, 2, 130, 130, True)])
def test_flash_attention(eng, orc, hd, Hq, Hkv, Tq, Tk, causal):
    import ctypes as C
    rng = np.random.default_rng(hd + Tq)
    B = 2
    q = bf(rng.standard_normal((B, Tq, Hq, hd))); k = bf(rng.standard_normal((B, Tk, Hkv, hd))); v = bf(rng.standard_normal((B, Tk, Hkv, hd)))
    got = eng.test_attention(q, k, v, causal)
    ref = np.empty_like(q)
    for b in range(B):
        o = np.empty((Tq, Hq, hd), np.float32)
        orc.lib().oracle_attention(q[b].ctypes.data, k[b].ctypes.data, v[b].ctypes.data, o.ctypes.data, Tq, Tk, Hq, Hkv, hd, int(causal), 0, 1)
        ref[b] = o
    err = np.abs(got - ref)
    assert err.max() <= 0.03 and err.mean() < 2e-3, (err.max(), err.mean())


def test_flash_attention_spike(eng, orc):
    # force the online-softmax rescale: one key row dominates late in the sequence (cdna_hip_programming.md rule 26)
    rng = np.random.default_rng(3)
    B, T, H, hd = 1, 300, 1, 64
    q = bf(rng.standard_normal((B, T, H, hd)) * 0.3); k = bf(rng.standard_normal((B, T, H, hd)) * 0.3); v = bf(rng.standard_normal((B, T, H, hd)))
    k[0, 257, 0] = bf(q[0, 10, 0] * 40.0)
    got = eng.test_attention(q, k, v, False)
    o = np.empty((T, H, hd), np.float32)
    orc.lib().oracle_attention(q[0].ctypes.data, k[0].ctypes.data, v[0].ctypes.data, o.ctypes.data, T, T, H, H, hd, 0, 0, 1)
    assert np.abs(got[0] - o).max() <= 0.03


def test_decode_attention(eng, orc):
    rng = np.random.default_rng(9)
    B, Tk, Hq, Hkv = 3, 277, 4, 1
    q = bf(rng.standard_normal((B, Hq, 128))); k = bf(rng.standard_normal((B, Tk, Hkv, 128))); v = bf(rng.standard_normal((B, Tk, Hkv, 128)))
    got = eng.test_decode_attention(q, k, v)
    for b in range(B):
        o = np.empty((1, Hq, 128), np.float32)
        qq = np.ascontiguousarray(q[b][None])
        orc.lib().oracle_attention(qq.ctypes.data, k[b].ctypes.data, v[b].ctypes.data, o.ctypes.data, 1, Tk, Hq, Hkv, 128, 0, Tk - 1, 1)
        assert np.abs(got[b] - o[0]).max() <= 0.02


def test_norms(eng, orc):
    rng = np.random.default_rng(11)
    for d in (128, 256, 1280, 2048):
        x = bf(rng.standard_normal((37, d)) * 2 + 0.3); w = bf(1 + 0.1 * rng.standard_normal(d)); b = bf(0.1 * rng.standard_normal(d))
        ref = np.empty_like(x)
        orc.lib().oracle_layernorm(x.ctypes.data, w.ctypes.data, b.ctypes.data, ref.ctypes.data, 37, d, 1e-5, 1)
        got = eng.test_layernorm(x, w, b, 1e-5, rms=False)
        assert np.all(np.abs(got - ref) <= ulp_tol(ref, 2))
        orc.lib().oracle_rmsnorm(x.ctypes.data, w.ctypes.data, ref.ctypes.data, 37, d, 1e-5, 1)
        got = eng.test_layernorm(x, w, None, 1e-5, rms=True)
        assert np.all(np.abs(got - ref) <= ulp_tol(ref, 2))


# ------------------------------------------------------------------------------------------ log-mel
MEL_TAGS = ["5s", "20s", "30s", "partial", "ragged", "short", "one"]


def test_logmel_vs_golden_and_oracle(eng, orc, golden_dir):
    segs, gold = [], []
    for tag in MEL_TAGS:
        g = np.load(os.path.join(golden_dir, f"mel_{tag}.npz"))
        segs.append(synth.synth_pcm(int(g["seg_index"]), int(g["n_samples"])))
        gold.append(g)
    segs.append(np.zeros(16000, np.int16)); segs.append(np.zeros(0, np.int16))
    feats, mask = eng.logmel(segs)
    for i, g in enumerate(gold):
        np.testing.assert_allclose(feats[i][:, ::7], g["feats_sub"], atol=1e-3, rtol=0)      # north_star tolerance
        np.testing.assert_allclose(feats[i][:, -16:], g["feats_tail"], atol=1e-3, rtol=0)
        assert int(mask[i].sum()) == int(g["mask_sum"])
        ref, m2 = orc.logmel(segs[i])
        assert np.abs(feats[i] - ref).max() < 1e-3
        assert np.array_equal(mask[i], m2)
    g = np.load(os.path.join(golden_dir, "mel_silence.npz"))
    np.testing.assert_allclose(feats[len(gold)][:, ::7], g["feats_sub"], atol=1e-5)
    assert np.allclose(feats[len(gold) + 1], -1.5, atol=1e-5) and mask[len(gold) + 1].sum() == 0   # empty input


def test_logmel_length_sweep_vs_oracle(eng, orc):
    """Segment lengths on and around every boundary the kernel cares about: hop (160), window (400), the 32-frame block (5120), the
    reflect padding, the 30 s cap - plus full-scale and alternating-sign inputs (the folded DFT adds and subtracts mirrored samples)."""
    lens = [1, 2, 159, 160, 161, 199, 200, 201, 399, 400, 401, 5119, 5120, 5121, 5279, 31999, 32000, 479839, 479999, 480000]
    segs = [synth.synth_pcm(300 + i, n) for i, n in enumerate(lens)]
    sq = np.full(20000, 32767, np.int16); sq[1::2] = -32768
    segs.append(sq)                                               # Nyquist square wave at full scale
    segs.append(np.full(7777, -32768, np.int16))                  # DC at full scale
    for lo in range(0, len(segs), 8):
        part = segs[lo:lo + 8]
        feats, mask = eng.logmel(part)
        for i, pcm in enumerate(part):
            ref, m2 = orc.logmel(pcm)
            assert np.array_equal(mask[i], m2), len(pcm)
            assert np.abs(feats[i] - ref).max() < 1e-3, (len(pcm), float(np.abs(feats[i] - ref).max()))


# ------------------------------------------------------------------------------------------ encoder + full path (TINY)
def _golden_case(golden_dir, si, tag="bf16"):
    g = np.load(os.path.join(golden_dir, f"tiny_{tag}.npz"))
    p = f"s{si}_"
    return g, p, synth.synth_pcm(int(g[p + "seg_index"]), int(g[p + "n_samples"]))


def test_encoder_vs_oracle_and_golden(eng, orc, golden_dir):
    d = spec.TINY
    state = synth.synth_state_dict(d, 20260128, bf16=True)
    om = orc.Model(d, state, bf16=True)
    for si in range(2):
        g, p, pcm = _golden_case(golden_dir, si)
        feats, mask = orc.logmel(pcm)
        emb, n_audio, layers, enc_out = eng.encode(feats[None], [int(mask.sum())], want_layers=True, want_enc_out=True)
        r = om.transcribe(feats, int(mask.sum()), g[p + "prompt_ids"], 1, want=("enc_layers", "enc_out"))
        assert int(n_audio[0]) == int(g[p + "n_audio"])
        for li in range(d.enc_layers):
            assert_bf16_close(layers[0, li], r["enc_layers"][li], f"encoder layer {li}")
        assert_bf16_close(emb[0, : n_audio[0]], r["audio_embeds"][: n_audio[0]], "audio embeds", min_exact=0.2)
        # and against the reference fixtures themselves (torch's own summation order: a little further than the oracle)
        assert_bf16_close(emb[0, : n_audio[0]], g[p + "audio_embeds"], "audio embeds vs reference", n_ulp_of_max=3.0, min_exact=0.15)
        assert_bf16_close(enc_out[0][::31], g[p + "enc_out_sub"], "encoder output vs reference", n_ulp_of_max=3.0, min_exact=0.3)


def test_transcribe_vs_golden(eng, golden_dir):
    """End to end through the hot call.  Logit bound: 4 bf16 ulp at |logit| < 4 (= 4 * 2^-6); token IDs bit-exact up to the first
    step whose reference margin is below twice that bound."""
    g = np.load(os.path.join(golden_dir, "tiny_bf16.npz"))
    n_new = int(g["n_new"])
    segs, prompts = [], []
    for si in range(2):
        p = f"s{si}_"
        segs.append(synth.synth_pcm(int(g[p + "seg_index"]), int(g[p + "n_samples"])))
        prompts.append(g[p + "prompt_ids"])
    ids, logits = eng.transcribe_batch(segs, prompts, [n_new, n_new], want_logits=True)
    tol = 4 * 2.0 ** -6
    for si in range(2):
        p = f"s{si}_"
        ref_ids, ref_logits, margins = g[p + "new_ids"], g[p + "step_logits"], g[p + "margins"]
        safe = margins > 2 * tol
        n_safe = len(ref_ids) if safe.all() else int(np.argmin(safe))
        assert n_safe >= 1
        assert np.array_equal(ids[si][:n_safe], ref_ids[:n_safe]), (ids[si], ref_ids)
        same = int(np.argmin(ids[si][: len(ref_ids)] == ref_ids)) if not np.array_equal(ids[si][: len(ref_ids)], ref_ids) else len(ref_ids)
        for s in range(min(same + 1, len(ref_ids))):   # logits are comparable while the generated prefix is identical
            dmax = float(np.abs(logits[s, si] - ref_logits[s]).max())
            if dmax > tol:   # diagnostics for a rare flake seen on some boxes
                from oracle import oracle as _orc
                allsteps = np.abs(logits[: len(ref_ids), si] - ref_logits).max(axis=1)
                ids2, logits2 = eng.transcribe_batch(segs, prompts, [n_new, n_new], want_logits=True)
                rerun = float(np.abs(logits2 - logits).max())
                rerun_ok = float(np.abs(logits2[: len(ref_ids), si] - ref_logits).max())
                fgpu, _ = eng.logmel(segs)
                fref = [_orc.logmel(x)[0] for x in segs]
                dmel = [float(np.abs(fgpu[i] - fref[i]).max()) for i in range(2)]
                emb, n_a, _, _ = eng.encode(np.stack(fref), [500, 2000])
                demb = [float(np.abs(emb[i, : n_a[i]] - g[f"s{i}_audio_embeds"]).max()) for i in range(2)]
                # localise inside the decoder prefill: hidden states after the embedding merge and after each layer vs the oracle
                om = _orc.Model(spec.TINY, synth.synth_state_dict(spec.TINY, 20260128, bf16=True), bf16=True)
                eng.set_option("prefill_taps", 1)
                ids3, logits3 = eng.transcribe_batch(segs, prompts, [n_new, n_new], want_logits=True)
                eng.set_option("prefill_taps", 0)
                P0 = len(prompts[0]); P1 = len(prompts[1]); dd = spec.TINY.dec_d
                r0 = om.transcribe(fref[0], 500, prompts[0], 1, want=("dec_layers",))
                tapinfo = []
                for li in range(spec.TINY.dec_layers + 1):
                    tp = eng.debug_read("prefill_tap", (P0 + P1, dd), li)
                    if li == 0:
                        tapinfo.append(("emb", float(np.abs(tp[:P0]).max()), float(np.abs(tp[4:4 + n_a[0]] - g["s0_audio_embeds"]).max())))
                    else:
                        e0 = np.abs(tp[:P0] - r0["dec_layers"][li - 1])
                        tapinfo.append((f"L{li - 1}", float(e0.max()), int(np.argmax(e0.max(axis=1))), np.round(e0.max(axis=1)[:8], 3).tolist()))
                emb1, n_a1, _, _ = eng.encode(fref[0][None], [500])
                demb1 = float(np.abs(emb1[0, : n_a1[0]] - g["s0_audio_embeds"]).max())
                raise AssertionError(f"seg {si} step {s}: max|dlogit| {dmax:.4f} > {tol}; per-step {np.round(allsteps[:6], 3).tolist()}; rerun-vs-first {rerun:.4f}, "
                                     f"rerun-vs-golden {rerun_ok:.4f}; logmel diff per seg {dmel}; embeds(B=2) diff {demb}; embeds(B=1, seg0) diff {demb1:.4f}; taps {tapinfo}; taps-run-vs-first {float(np.abs(logits3 - logits).max()):.4f}")


def test_random_cases_vs_oracle(eng, orc):
    """Seeded random sweep at TINY dimensions: segment lengths from 0.2 s to 30 s, prompt prefixes / suffixes of different lengths,
    different budgets, mixed in batches of 6 - every row against the bf16 oracle (token IDs while the reference margin allows, logits
    while the histories agree)."""
    d = spec.TINY
    om = orc.Model(d, synth.synth_state_dict(d, 20260128, bf16=True), bf16=True)
    rng = np.random.default_rng(77)
    tol = 4 * 2.0 ** -6
    checked = 0
    for batch in range(3):
        lens = [int(rng.integers(3200, 480000)) for _ in range(6)]
        segs = [synth.synth_pcm(500 + batch * 6 + i, n) for i, n in enumerate(lens)]
        prompts, budgets = [], []
        for n in lens:
            pre = [1] + [int(x) for x in rng.integers(2, 900, int(rng.integers(1, 6)))]
            suf = [int(x) for x in rng.integers(2, 900, int(rng.integers(1, 8)))]
            prompts.append(pre + [d.audio_token_id] * spec.audio_token_count(spec.valid_frames(n)) + suf)
            budgets.append(int(rng.integers(1, 9)))
        ids, logits = eng.transcribe_batch(segs, prompts, budgets, want_logits=True)
        for i, pcm in enumerate(segs):
            feats, mask = orc.logmel(pcm)
            r = om.transcribe(feats, int(mask.sum()), prompts[i], budgets[i])
            ref_ids, ref_logits = np.asarray(r["new_ids"]), np.asarray(r["step_logits"])
            assert len(ids[i]) == budgets[i]
            for st in range(budgets[i]):
                srt = np.sort(ref_logits[st]); margin = srt[-1] - srt[-2]
                assert np.abs(logits[st, i] - ref_logits[st]).max() <= tol, (batch, i, st)
                checked += 1
                if margin <= 2 * tol or ids[i][st] != ref_ids[st]:
                    assert margin <= 2 * tol, (batch, i, st, ids[i], ref_ids)      # a flip is only legitimate on a near-tie
                    break
    assert checked >= 30


def test_batch_matches_single_and_graph_matches_eager(eng, golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny_bf16.npz"))
    segs = [synth.synth_pcm(20 + i, n) for i, n in enumerate((80000, 320000, 20480, 123457))]
    prompts = []
    for s in segs:
        n_audio = spec.audio_token_count(spec.valid_frames(len(s)))
        prompts.append([1, 17, 23, 5] + [spec.TINY.audio_token_id] * n_audio + [7, 301, 302, 303, 9, 11])
    mn = [12, 20, 6, 9]
    batch_ids, _ = eng.transcribe_batch(segs, prompts, mn)                       # hipGraph decode loop
    eager_ids, _ = eng.transcribe_batch(segs, prompts, mn, want_logits=True)      # eager decode loop
    for i in range(4):
        assert len(batch_ids[i]) == mn[i]
        assert np.array_equal(batch_ids[i], eager_ids[i])
        single, _ = eng.transcribe_batch([segs[i]], [prompts[i]], [mn[i]])
        assert np.array_equal(single[0], batch_ids[i])                              # per-segment result independent of batching


def test_eos_stops_row(eng):
    # make the model's favourite token an EOS: run once, then rebuild with that id as EOS
    from dataclasses import replace
    from sonicscribe_amd.engine import Engine
    seg = synth.synth_pcm(31, 80000)
    n_audio = spec.audio_token_count(spec.valid_frames(len(seg)))
    prompt = [1, 17, 23, 5] + [spec.TINY.audio_token_id] * n_audio + [7, 301, 302, 303, 9, 11]
    ids, _ = eng.transcribe_batch([seg], [prompt], [8])
    d2 = replace(spec.TINY, eos_ids=(int(ids[0][2]), 991, 992))
    e2 = Engine(d2, 0, max_batch=2, max_ctx=512)
    e2.load_synthetic(20260128)
    ids2, _ = e2.transcribe_batch([seg, seg], [prompt, prompt], [8, 3])
    first = int(np.argmax(ids[0] == ids[0][2]))
    assert np.array_equal(ids2[0], ids[0][: first + 1])      # stops right after emitting EOS
    assert len(ids2[1]) <= 3
    e2.close()


def test_mismatch_and_limits_raise(eng):
    seg = synth.synth_pcm(1, 80000)
    with pytest.raises(ValueError):
        eng.transcribe_batch([seg], [[1, spec.TINY.audio_token_id, 2]], [4])
    with pytest.raises(RuntimeError):
        eng.transcribe_batch([seg], [[1] * 600], [4])      # exceeds max_ctx


def test_multi_window_request(eng):
    # 35 s audio -> two windows in one request (processing_glmasr.py:136-157)
    from sonicscribe_amd import frontend
    pcm = synth.synth_pcm(40, 560000)
    wins = [pcm[s:e] for s, e in frontend.split_windows(len(pcm), spec.TINY)]
    total, per_win = frontend.request_audio_tokens(len(pcm), spec.TINY)
    assert len(wins) == 2
    prompt = [1] + [spec.TINY.audio_token_id] * sum(per_win) + [7]
    ids, _ = eng.transcribe_batch(wins, [prompt], [5], req_win=[0, 2])
    assert len(ids[0]) == 5


def test_asrmodel_facade_threads(golden_dir):
    import threading
    from sonicscribe_amd.asr import ASRModel
    m = ASRModel.from_synthetic(spec.TINY, max_batch=8, max_ctx=512)
    segs = [synth.synth_pcm(50 + i, 80000).astype(np.float32) / 32768.0 for i in range(4)]
    want = [m.transcribe(s[None], 16000, max_new_tokens=6) for s in segs]
    got = [None] * 4

    def work(i):
        got[i] = m.transcribe(segs[i][None], sampling_rate=16000, max_new_tokens=6)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert got == want and all(isinstance(x, str) and x for x in got)
    info = m.transcribe(segs[0][None], 16000, 4, hotwords=["Alpha", "alpha ", "beta"], return_debug_info=True)
    assert set(info) >= {"transcript", "processing_time", "audio_length_sec", "mode", "device"}
    assert abs(info["audio_length_sec"] - 5.0) < 1e-6
    assert m.get_model_info()["mode"] == "native"
    assert hasattr(m, "model")
    del m.model                                             # main.py:84-86
    assert not hasattr(m, "model")
    with pytest.raises(RuntimeError):
        m.transcribe(segs[0][None])


def test_fullwidth_layer_vs_oracle(orc):
    """Full-width GLM-ASR-Nano dimensions (d=1280/5120, 20 heads; decoder 2048/6144, GQA 16:4) with ONE encoder and ONE decoder
    layer and a small vocab, two 20 s segments: drives every large-shape kernel instantiation (256x256 GEMM with the
    QKV+V^T / GELU / residual / SwiGLU epilogues, T=1500 flash attention, prefill at M=520) against the bf16 oracle."""
    from dataclasses import replace
    from sonicscribe_amd.engine import Engine
    d = replace(spec.FULL, enc_layers=1, dec_layers=1, vocab=1024, audio_token_id=1000, eos_ids=(990, 991, 992))
    seed = 7
    e = Engine(d, 0, max_batch=2, max_ctx=320)
    e.load_synthetic(seed)
    om = orc.Model(d, synth.synth_state_dict(d, seed, bf16=True), bf16=True)
    segs = [synth.synth_pcm(60 + i, 320000) for i in range(2)]
    n_audio = spec.audio_token_count(spec.valid_frames(320000))
    prompt = [1, 17, 23, 5] + [d.audio_token_id] * n_audio + [7, 301, 302, 303, 9, 11]
    n_new = 3
    ids, logits = e.transcribe_batch(segs, [prompt, prompt], [n_new, n_new], want_logits=True)
    feats_all, masks = e.logmel(segs)
    emb, n_a, layers, enc_out = e.encode(feats_all, [int(m.sum()) for m in masks], want_layers=True, want_enc_out=True)
    for i in range(2):
        feats, mask = orc.logmel(segs[i])
        r = om.transcribe(feats, int(mask.sum()), prompt, n_new, want=("enc_layers", "enc_out"))
        assert_bf16_close(layers[i, 0], r["enc_layers"][0], "full-width encoder layer")
        assert_bf16_close(emb[i, :n_audio], r["audio_embeds"][:n_audio], "full-width audio embeds", min_exact=0.2)
        for st in range(n_new):                                # step 0 = prefill; steps 1.. = decode kernels (fused o_proj/residual/RMSNorm/gate-up)
            if st and ids[i][st - 1] != r["new_ids"][st - 1]:
                break                                          # histories diverged on a near-tie: later steps are not comparable
            dl = np.abs(logits[st, i] - r["step_logits"][st])
            assert dl.max() <= 6 * 2.0 ** -6, (st, dl.max())   # bf16-derived bound
        srt = np.sort(r["step_logits"][0]); margin = srt[-1] - srt[-2]
        if margin > 12 * 2.0 ** -6:
            assert ids[i][0] == r["new_ids"][0]
    # the unfused decode path (o_proj slabs -> add+RMSNorm kernel -> gate/up) must agree with the fused one to bf16 noise
    e.set_option("no_fused_gu", 1)
    try:
        ids_u, logits_u = e.transcribe_batch(segs, [prompt, prompt], [n_new, n_new], want_logits=True)
    finally:
        e.set_option("no_fused_gu", 0)
    for i in range(2):
        if np.array_equal(ids_u[i], ids[i]):
            assert np.abs(logits_u[:, i] - logits[:, i]).max() <= 4 * 2.0 ** -6
    e.close()


@pytest.mark.parametrize("R", [5, 17, 32, 33, 47, 64])
def test_fused_decode_rows_vs_unfused(R):
    """Fused decode kernels (o_proj + residual + partials, gate/up with in-LDS RMSNorm) at ragged row counts: full decoder width,
    two layers, R sequences of different prompt lengths, against the slab + add/RMSNorm path that the skinny-GEMM tests pin."""
    from dataclasses import replace
    from sonicscribe_amd.engine import Engine
    d = replace(spec.FULL, enc_layers=1, dec_layers=2, vocab=1024, audio_token_id=1000, eos_ids=())
    e = Engine(d, 0, max_batch=64, max_ctx=384)
    e.load_synthetic(11)
    lens = [16000 * (1 + (i % 5)) + 37 * i for i in range(R)]                 # 1..5 s: 12..62 audio tokens
    segs = [synth.synth_pcm(400 + i, n) for i, n in enumerate(lens)]
    prompts = [[1, 17, 23, 5] + [d.audio_token_id] * spec.audio_token_count(spec.valid_frames(n)) + [7, 301, 302, 303, 9, 11][: 3 + i % 4]
               for i, n in enumerate(lens)]
    n_new = 6
    ids_f, log_f = e.transcribe_batch(segs, prompts, [n_new] * R, want_logits=True)
    ids_g, _ = e.transcribe_batch(segs, prompts, [n_new] * R)                  # graph replay of the same fused path
    e.set_option("no_fused_gu", 1)
    try:
        ids_u, log_u = e.transcribe_batch(segs, prompts, [n_new] * R, want_logits=True)
    finally:
        e.set_option("no_fused_gu", 0)
    if R > 32:
        # round 5: 33 .. 64 rows go through skinny_gu64_kernel (one sweep, four 16-row passes); round 4's two passes of 32 rows must give the same BITS
        # (and the fused o_proj's 32-row blocks the same bits as round 4's 16-row ones)
        # ... and the split form continuous loops use (RMSNorm by its own kernel from the same partials + gate/up without the in-LDS norm)
        for opt in ("gu64_two_pass", "o64_16rows", "gu64_split_norm"):
            e.set_option(opt, 1)
            try:
                ids_2, log_2 = e.transcribe_batch(segs, prompts, [n_new] * R, want_logits=True)
            finally:
                e.set_option(opt, 0)
            assert all(np.array_equal(ids_f[i], ids_2[i]) for i in range(R)) and np.array_equal(log_f.view(np.uint32), log_2.view(np.uint32)), opt
    same = 0
    for i in range(R):
        assert np.array_equal(ids_f[i], ids_g[i])
        if np.array_equal(ids_f[i], ids_u[i]):
            same += 1
            assert np.abs(log_f[:, i] - log_u[:, i]).max() <= 4 * 2.0 ** -6, i
    assert same >= R - 1 - R // 32          # a near-tie may flip one history (two beyond 32 rows) between the two summation orders, not more
    # batch invariance of the fused path, 33 .. 64 rows included (round 4: the two-pass gate/up kernel gives every row the arithmetic of the
    # one-pass kernel): rows alone give the same BITS as inside the batch
    for i in sorted({0, R // 2, R - 1}):
        ids_s, log_s = e.transcribe_batch([segs[i]], [prompts[i]], [n_new], want_logits=True)
        assert np.array_equal(ids_s[0], ids_f[i]) and np.array_equal(log_s[:, 0].view(np.uint32), log_f[:, i].view(np.uint32)), (R, i)
    e.close()


def test_checkpoint_loader_equals_synthetic(tmp_path):
    """ASRModel(checkpoint_dir): config.json + bf16 safetensors in the on-disk HF layout -> same tokens as the device-side generator."""
    from sonicscribe_amd import weights
    from sonicscribe_amd.asr import ASRModel
    seed = 11
    weights.save_synthetic_checkpoint(str(tmp_path), spec.TINY, seed)
    m1 = ASRModel(str(tmp_path), device="cuda", mode="native", max_batch=2, max_ctx=512, _allow_synthetic_prompt=True)
    m2 = ASRModel.from_synthetic(spec.TINY, seed=seed, max_batch=2, max_ctx=512)
    assert m1.dims == spec.TINY
    wav = synth.synth_pcm(70, 80000).astype(np.float32) / 32768.0
    t1 = m1.transcribe(wav[None], 16000, max_new_tokens=10)
    t2 = m2.transcribe(wav[None], 16000, max_new_tokens=10)
    assert t1 == t2 and len(t1.split()) == 10
    assert m1.transcribe_batch([wav, wav[:40000]], max_new_tokens=[4, 6])[0] == " ".join(t1.split()[:4])
    m1.close(); m2.close()


def test_streaming_sessions_coalesced():
    """Config 5's call pattern (transcription_manager.py:19-65): every session turns int16 chunk bytes into a float [1,N] tensor and
    calls transcribe() -- partials on the last 20 chunks (1.28 s, 15 tokens), finals on the whole segment with
    min(50 + 5*duration, 200) tokens -- from concurrent threads.  Results must equal the one-at-a-time results."""
    import threading
    from sonicscribe_amd import frontend
    from sonicscribe_amd.asr import ASRModel
    m = ASRModel.from_synthetic(spec.TINY, max_batch=16, max_ctx=512)
    chunk = 2048                                                   # AUDIO_CHUNK_SIZE bytes (config.py:24)
    sessions = []
    for sidx in range(6):
        pcm = synth.synth_pcm(80 + sidx, 16000 * (3 + sidx))     # 3..8 s of speech
        data = pcm.tobytes()
        n_chunks = len(data) // chunk
        partial = data[(n_chunks - 20) * chunk: n_chunks * chunk]  # audio_manager.py:106-114: last <= 20 chunks
        sessions.append((partial, data, len(pcm) / 16000.0))

    def call(audio_bytes, max_new):
        t = frontend.pcm_bytes_to_float(audio_bytes)               # transcription_manager.py:45-54
        return m.transcribe(t, sampling_rate=16000, max_new_tokens=max_new).strip()

    want = [(call(p, 15), call(f, min(frontend.max_new_tokens_committed(d), 40))) for p, f, d in sessions]
    got = [[None, None] for _ in sessions]

    def worker(i, which):
        p, f, d = sessions[i]
        got[i][which] = call(p, 15) if which == 0 else call(f, min(frontend.max_new_tokens_committed(d), 40))
    ts = [threading.Thread(target=worker, args=(i, w)) for i in range(len(sessions)) for w in (0, 1)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert [tuple(x) for x in got] == want
    assert all(len(a.split()) == 15 for a, _ in want)
    m.close()


@pytest.mark.parametrize("dims_tag", ["fullwidth", "tiny"])
def test_two_row_decode_step_without_add_rmsnorm_launches(dims_tag):
    """Round 6: at <= 2 rows the token step has five launches per layer - the q|k|v projection (and, behind the last layer, the
    lm_head) sums down_proj's slabs, adds the residual and normalises its rows itself (skinny_xs_kernel<.., PRE>).  It restates add_rmsnorm_kernel
    statement by statement, so: the logits of every step are the same BITS as with the standalone launches (option no_pre_norm), eagerly and through
    the captured graphs, a row alone equals the same row inside a batch of 5 (which takes the six-launch chain), and a continuous loop with one or
    two occupied rows (the R = 2 chunk graph) returns the solo tokens.  HF semantics: modeling_llama.py:60-65, 306-324."""
    from dataclasses import replace
    from sonicscribe_amd.engine import Engine
    d = (replace(spec.FULL, enc_layers=1, dec_layers=3, vocab=1024, audio_token_id=1000, eos_ids=()) if dims_tag == "fullwidth"
         else replace(spec.TINY, eos_ids=()))
    e = Engine(d, 0, max_batch=16, max_ctx=384)
    e.load_synthetic(11)
    lens = [16000 * (1 + i) + 37 * i for i in range(5)]
    segs = [synth.synth_pcm(700 + i, n) for i, n in enumerate(lens)]
    prompts = [[1, 17, 23, 5] + [d.audio_token_id] * spec.audio_token_count(spec.valid_frames(n)) + [7, 301, 302, 303, 9, 11][: 3 + i % 4] for i, n in enumerate(lens)]
    n_new = 9
    ids5, log5 = e.transcribe_batch(segs, prompts, [n_new] * 5, want_logits=True)                  # five rows: the six-launch chain
    per_layer_5 = e.timings()["decode_launches_per_layer"]
    for R in (1, 2):
        ids_p, log_p = e.transcribe_batch(segs[:R], prompts[:R], [n_new] * R, want_logits=True)
        took = e.timings()["decode_launches_per_layer"]
        ids_g, _ = e.transcribe_batch(segs[:R], prompts[:R], [n_new] * R)                         # graph replay
        e.set_option("no_pre_norm", 1)
        try:
            ids_n, log_n = e.transcribe_batch(segs[:R], prompts[:R], [n_new] * R, want_logits=True)
            took_n = e.timings()["decode_launches_per_layer"]
        finally:
            e.set_option("no_pre_norm", 0)
        assert np.array_equal(log_p.view(np.uint32), log_n.view(np.uint32)), R
        assert np.array_equal(log_p.view(np.uint32), log5[:, :R].view(np.uint32)), R             # batch invariance across the two chains
        for i in range(R):
            assert np.array_equal(ids_p[i], ids_n[i]) and np.array_equal(ids_p[i], ids_g[i]) and np.array_equal(ids_p[i], ids5[i])
        if dims_tag == "fullwidth":
            assert (took, took_n, per_layer_5) == (5, 6, 6), (took, took_n, per_layer_5)          # the five-launch chain really ran
    # continuous loop: one, then two occupied rows (R = 2 chunk graphs), then a third joins (R = 16 graphs)
    pre = e.slot()
    e.service_begin()
    got = {}
    pre.stage_pcm(segs[:1]); pre.prefill(prompts[:1], [n_new])
    seq = {0: e.splice_rows(pre, [0], [0])}
    for step in range(200):
        if step == 1:
            pre.stage_pcm(segs[1:2]); pre.prefill(prompts[1:2], [n_new]); seq[1] = e.splice_rows(pre, [0], [1])
        if step == 2:
            pre.stage_pcm(segs[2:3]); pre.prefill(prompts[2:3], [n_new]); seq[2] = e.splice_rows(pre, [0], [2])
        top = max([r for r in seq if r not in got], default=-1) + 1
        if top == 0 and step > 2:
            break
        fin, nn, s_, _ = e.service_step(1, max(top, 1))
        for r in list(seq):
            if r not in got and s_ > seq[r] and fin[r]:
                got[r] = e.fetch_row(r, int(nn[r]))
    e.service_end()
    for r in range(3):
        assert np.array_equal(got[r], ids5[r]), r
    e.close()
