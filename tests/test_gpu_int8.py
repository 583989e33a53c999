"""GPU tests of the INT8 path (BASELINE config 4): backend/asr.py mode="int8" = fp16 activations + bitsandbytes
LLM.int8() linears (asr.py:148-210).  bitsandbytes is absent offline, so the checker is the restatement in oracle/sonic_oracle.c
(linear_int8, MODE_INT8) -- PARITY UNPINNED against the real library, pinned only to its published algorithm.

What is exact and what is not: the int8 x int8 -> int32 products are exact on both sides, and the dequantisation follows the same
operation order, so a Linear8bitLt on identical inputs must agree bit for bit.  End to end, the fp16 ops between the linears
(LayerNorm, attention, GELU) differ in fp32 summation order; a 1-ulp fp16 difference on an activation can move an int8 code by one,
so whole-model logits are compared with a stated bound.
"""
import os
from dataclasses import replace

import numpy as np
import pytest

from sonicscribe_amd import spec, synth

pytestmark = pytest.mark.gpu
SEED = 20260128


def f16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


@pytest.fixture(scope="module")
def eng8():
    from sonicscribe_amd.engine import Engine, MODE_INT8
    e = Engine(spec.TINY, 0, MODE_INT8, max_batch=8, max_ctx=512)
    e.load_synthetic(SEED)
    yield e
    e.close()


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def _ref_linear(orc, X, W, bias, group_rows):
    cb, scb = orc.quantize_rows(W)
    out = np.empty((X.shape[0], W.shape[0]), np.float32)
    for r0 in range(0, X.shape[0], group_rows):
        out[r0:r0 + group_rows] = orc.linear_int8(X[r0:r0 + group_rows], cb, scb, bias)
    return out


@pytest.mark.parametrize("M,N,K,group_rows", [(16, 64, 128, 16), (100, 128, 256, 25), (300, 256, 384, 300), (777, 384, 512, 259), (1536, 512, 1280, 512),
                                              (2048, 256, 5120, 1024), (600, 1280, 512, 200)])
def test_linear8bit_bias_exact(eng8, orc, M, N, K, group_rows):
    """One Linear8bitLt + bias against the oracle: must be BIT-EXACT (integer products, same dequantisation order).  Shapes cover the
    128x128 and the 256x256 int8 MFMA kernels (M >= 512, N >= 256, K >= 512), ragged edges, several groups per call, and planted
    outliers: single elements >= 6 (their column leaves the int8 product for the whole group) in some groups only."""
    rng = np.random.default_rng(M + N + K)
    X = f16(rng.standard_normal((M, K)) * 1.2); W = f16(rng.standard_normal((N, K)) * 0.06); b = f16(rng.standard_normal(N) * 0.1)
    X[0, 3] = 6.0; X[M // 2, K - 1] = -11.5; X[M - 1, 64] = 7.25; X[M // 2, 17] = 5.99609375  # 6.0 itself is an outlier, the fp16 below it is not
    got = eng8.test_linear_int8(X, W, b, group_rows=group_rows)
    ref = _ref_linear(orc, X, W, b, group_rows)
    assert np.array_equal(got, ref), (np.abs(got - ref).max(), int((got != ref).sum()))
    dense = X.astype(np.float64) @ W.T.astype(np.float64) + b
    assert np.abs(got - dense).max() < 0.02 * np.sqrt(K)                                      # and it is a sane approximation of the fp product (int8 noise ~ sqrt(K))


def test_linear8bit_int8_mfma_layout(eng8, orc):
    """Exact-integer data with an asymmetric weight matrix: a wrong lane / k map of v_mfma_i32_16x16x64_i8 cannot hide (cdna guide §3)."""
    M, N, K = 64, 64, 256
    X = np.zeros((M, K), np.float32); W = np.zeros((N, K), np.float32)
    for m in range(M):
        X[m, (m * 7) % K] = 1.0; X[m, (m * 3 + 1) % K] = 0.5
    W = f16(((np.arange(N)[:, None] * 31 + np.arange(K)[None, :] * 17) % 127 - 63) / 64.0)
    got = eng8.test_linear_int8(X, W, None)
    ref = _ref_linear(orc, X, W, None, M)
    assert np.array_equal(got, ref)


def test_linear8bit_epilogues(eng8, orc):
    import math
    from sonicscribe_amd.engine import EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_SWIGLU
    rng = np.random.default_rng(8)
    for (M, N, K) in [(200, 256, 256), (1024, 512, 768)]:
        X = f16(rng.standard_normal((M, K))); W = f16(rng.standard_normal((N, K)) * 0.08); b = f16(rng.standard_normal(N) * 0.1)
        X[5, 9] = 8.0
        R = f16(rng.standard_normal((M, N)))
        lin = _ref_linear(orc, X, W, b, M)
        got = eng8.test_linear_int8(X, W, b, resid=R, epi=EPI_BIAS_RESID)
        assert np.array_equal(got, f16(lin + R))
        erf = np.vectorize(math.erf)
        gelu = f16((0.5 * lin * (1.0 + erf(lin.astype(np.float64) / math.sqrt(2.0)))).astype(np.float32))
        got = eng8.test_linear_int8(X, W, b, epi=EPI_BIAS_GELU)
        assert np.abs(got - gelu).max() <= 2.0 ** -10 * max(1.0, float(np.abs(gelu).max()))    # 1 fp16 ulp (fast erf / exp in the epilogue)
        ff = N // 2
        Wg = W[:ff]; Wu = W[ff:]
        Wi = np.empty_like(W)
        for g in range(ff // 16):
            Wi[32 * g:32 * g + 16] = Wg[16 * g:16 * g + 16]; Wi[32 * g + 16:32 * g + 32] = Wu[16 * g:16 * g + 16]
        gg = _ref_linear(orc, X, Wg, None, M); uu = _ref_linear(orc, X, Wu, None, M)
        ref = f16(f16(gg / (1.0 + np.exp(-gg.astype(np.float64))).astype(np.float32)) * uu)
        got = eng8.test_linear_int8(X, Wi, None, epi=EPI_SWIGLU)
        assert np.abs(got - ref).max() <= 2.0 ** -9 * max(1.0, float(np.abs(ref).max()))


def test_linear8bit_long_outlier_lists_go_to_the_dense_side_product(eng8, orc):
    """Requests with more than 8 outlier columns (synthetic SwiGLU activations: ~360 of 6144) leave a residual-epilogue int8 GEMM without
    their outlier sum; a dense fp16 MFMA product over the gathered columns finishes them (bitsandbytes does the same:
    MatMul8bitLt.forward adds subA @ subB).  Two requests in one call, 260 rows each (a 64-row tile of the side kernel spans both):
    request 0 with 70 outlier columns (deferred), request 1 with 5 (walked in the GEMM epilogue, exact).  The MFMA sums in its own order, the
    oracle in ascending k: a deferred element may differ by one fp16 ulp when v + sum sits on a rounding boundary; with the deferral
    switched off the result is bit-exact again."""
    from sonicscribe_amd.engine import EPI_BIAS_RESID
    rng = np.random.default_rng(99)
    M, N, K, G = 520, 256, 512, 260
    X = f16(rng.standard_normal((M, K))); W = f16(rng.standard_normal((N, K)) * 0.08); b = f16(rng.standard_normal(N) * 0.1)
    R = f16(rng.standard_normal((M, N)))
    cols0 = rng.choice(K, 70, replace=False)
    X[rng.integers(0, G, 70), cols0] = f16(rng.choice([-1.0, 1.0], 70) * rng.uniform(6.0, 14.0, 70))
    cols1 = rng.choice(K, 5, replace=False)
    X[G + rng.integers(0, G, 5), cols1] = f16(rng.uniform(6.5, 9.0, 5))
    ref = f16(_ref_linear(orc, X, W, b, G) + R)
    got = eng8.test_linear_int8(X, W, b, resid=R, group_rows=G, epi=EPI_BIAS_RESID)
    lin = _ref_linear(orc, X, W, b, G)                              # the module output before the residual: the rounding that may move is ITS last one

    def ulp16(v):
        return np.maximum(2.0 ** (np.floor(np.log2(np.maximum(np.abs(v), 2.0 ** -14))) - 10), 2.0 ** -24)
    ulp = ulp16(lin) + ulp16(ref)                                   # one ulp of fp16(v + sum), carried through fp16(. + R)
    diff = np.abs(got - ref)
    assert (diff <= ulp).all(), float((diff / ulp).max())
    assert np.array_equal(got[G:], ref[G:])                         # the short list stayed on the exact path
    frac = float((diff[:G] > 0).mean())
    print(f"deferred request: {frac * 100:.3f} % of the elements differ from the ascending-k sum (by one fp16 ulp)")
    assert frac < 0.02
    eng8.set_option("i8_defer_thr", -1)
    try:
        exact = eng8.test_linear_int8(X, W, b, resid=R, group_rows=G, epi=EPI_BIAS_RESID)
    finally:
        eng8.set_option("i8_defer_thr", 8)
    assert np.array_equal(exact, ref)


def _prompt(n_samples, d):
    return [1, 17, 23, 5] + [d.audio_token_id] * spec.audio_token_count(spec.valid_frames(n_samples)) + [7, 301, 302, 303, 9, 11]


def test_int8_transcribe_vs_oracle(eng8, orc):
    """Whole path in int8 mode at TINY dims against the oracle's MODE_INT8: fp16 log-mel hand-off, fp16 conv stem / norms / attention,
    Linear8bitLt everywhere else, fp16 lm_head, greedy; free-running and teacher-forced, a 5 s and a 20 s segment in one batch."""
    d = spec.TINY
    om = orc.Model(d, synth.synth_state_dict(d, SEED, 2), mode=orc.MODE_INT8)
    segs = [synth.synth_pcm(10, 80000), synth.synth_pcm(11, 320000)]
    prompts = [_prompt(len(s), d) for s in segs]
    n_new = 8
    rng = np.random.default_rng(5)
    force = rng.integers(2, 900, (2, n_new)).astype(np.int32)
    tol = 0.1        # measured 0.073: a 1-ulp fp16 difference on an activation can move an int8 code by one (1/127 of the row's range)
    worst = 0.0
    for forced in (None, force):
        eng8.set_forced_ids(forced)
        try:
            ids, logits = eng8.transcribe_batch(segs, prompts, [n_new, n_new], want_logits=True)
        finally:
            eng8.set_forced_ids(None)
        for i in range(2):
            feats, mask = orc.logmel(segs[i])
            r = om.transcribe(feats, int(mask.sum()), prompts[i], n_new, force_ids=None if forced is None else forced[i])
            n = min(len(ids[i]), len(r["new_ids"]))
            for st in range(n):
                dl = float(np.abs(logits[st, i] - r["step_logits"][st]).max()); worst = max(worst, dl)
                assert dl <= tol, (forced is not None, i, st, dl)
                srt = np.sort(r["step_logits"][st]); margin = srt[-1] - srt[-2]
                if ids[i][st] != r["new_ids"][st]:
                    assert margin <= 2 * tol, (i, st, ids[i], r["new_ids"])
                    break
    print(f"int8 tiny: max|dlogit| vs oracle {worst:.4f}")
    # graph-replayed loop == eager loop, batch == single
    g, _ = eng8.transcribe_batch(segs, prompts, [n_new, n_new])
    ee, _ = eng8.transcribe_batch(segs, prompts, [n_new, n_new], want_logits=True)
    s0, _ = eng8.transcribe_batch([segs[1]], [prompts[1]], [n_new])
    assert np.array_equal(g[0], ee[0]) and np.array_equal(g[1], ee[1]) and np.array_equal(s0[0], g[1])


def test_int8_multi_window_request_vs_oracle(orc):
    """A 65 s request (three 30 s windows behind one prompt) in INT8 mode, batched with a short one: HF runs the windows of a request
    as one encoder batch, so LLM.int8 finds its outlier columns over ALL rows of the request (engine: group = request via the
    window -> request map) -- against the multi-window oracle, teacher-forced."""
    from sonicscribe_amd import frontend
    from sonicscribe_amd.engine import Engine, MODE_INT8
    d = spec.TINY
    eng8 = Engine(d, 0, MODE_INT8, max_batch=8, max_ctx=1024)
    eng8.load_synthetic(SEED)
    om = orc.Model(d, synth.synth_state_dict(d, SEED, 2), mode=orc.MODE_INT8)
    n_long, n_short = 65 * 16000, 3 * 16000
    pcm_l = frontend.normalise_to_int16(synth.synth_pcm(21, n_long).astype(np.float32) / 32768.0)
    pcm_s = frontend.normalise_to_int16(synth.synth_pcm(22, n_short).astype(np.float32) / 32768.0)
    wins = [pcm_l[s0:e0] for s0, e0 in frontend.split_windows(n_long, d)]
    assert len(wins) == 3
    n_audio, _ = frontend.request_audio_tokens(n_long, d)
    prompt_l = [1, 17, 23, 5] + [d.audio_token_id] * n_audio + [7, 301, 302, 303, 9, 11]
    prompt_s = _prompt(n_short, d)
    n_new = 5
    force = np.random.default_rng(9).integers(2, 900, (2, n_new)).astype(np.int32)
    eng8.set_forced_ids(force)
    try:
        ids, logits = eng8.transcribe_batch([pcm_s] + wins, [prompt_s, prompt_l], [n_new, n_new], req_win=[0, 1, 4], want_logits=True)
    finally:
        eng8.set_forced_ids(None)
    fm = [orc.logmel(w) for w in wins]
    ref = om.transcribe(np.stack([f for f, _ in fm]), [int(m.sum()) for _, m in fm], prompt_l, n_new, force_ids=force[1])
    worst = float(np.abs(logits[:, 1] - ref["step_logits"]).max())
    f0, m0 = orc.logmel(pcm_s)
    ref_s = om.transcribe(f0, int(m0.sum()), prompt_s, n_new, force_ids=force[0])
    worst = max(worst, float(np.abs(logits[:, 0] - ref_s["step_logits"]).max()))
    print(f"int8 multi-window: max|dlogit| vs oracle {worst:.4f}")
    eng8.close()
    assert worst <= 0.1


def test_int8_fullwidth_layer_vs_oracle(orc):
    """Full-width layers (encoder 1280 / 5120, decoder 2048 / 6144 GQA 16:4) at depth 1 + 1, vocabulary 1024, two 20 s segments:
    the 256x256 int8 GEMM at M = 3000 with every epilogue, the int8 decode-step kernels at their full-size tilings (K slices of 1024,
    down_proj with 6 slabs), against the oracle."""
    from sonicscribe_amd.engine import Engine, MODE_INT8
    d = replace(spec.FULL, enc_layers=1, dec_layers=1, vocab=1024, audio_token_id=1000, eos_ids=(990, 991, 992))
    seed = 7
    e = Engine(d, 0, MODE_INT8, max_batch=2, max_ctx=320)
    e.load_synthetic(seed)
    st = {}
    for name, shape, kind in spec.tensor_inventory(d):
        scale, offset = synth.kind_params(kind, shape)
        st[name] = orc.synth_fill(seed, name, int(np.prod(shape)), scale, offset, 2).reshape(shape)
    om = orc.Model(d, st, mode=orc.MODE_INT8)
    segs = [synth.synth_pcm(60 + i, 320000) for i in range(2)]
    prompt = _prompt(320000, d)
    n_new = 3
    force = np.asarray([[100, 200, 300], [400, 500, 600]], np.int32)
    e.set_forced_ids(force)
    ids, logits = e.transcribe_batch(segs, [prompt, prompt], [n_new, n_new], want_logits=True)
    e.set_forced_ids(None)
    worst = 0.0
    for i in range(2):
        feats, mask = orc.logmel(segs[i])
        r = om.transcribe(feats, int(mask.sum()), prompt, n_new, force_ids=force[i])
        worst = max(worst, float(np.abs(logits[:, i] - r["step_logits"]).max()))
    print(f"int8 full-width 1+1: max|dlogit| vs oracle {worst:.4f}")
    assert worst <= 0.1
    e.close()


@pytest.mark.parametrize("dims_name", ["tiny", "fullwidth"])
def test_int8_quantise_on_the_fly_equals_separate_pass(dims_name):
    """Decode step, int8 mode: o_proj's input is quantised by the projection kernel itself while it stages its X slice (row absmax gathered
    by the attention blocks with atomicMax, outliers listed by the consumer) instead of by a one-block-per-row launch in between
    (`i8_no_xq=1`, the round-2 form).  Same quantisation arithmetic, exact int32 sums, same outlier order: the step logits must be
    bit-identical, at the tiny tilings (K slices of 256) and at the full-size ones (K slices of 512, 5 rows so that the padded rows of the
    64-row image are exercised)."""
    from sonicscribe_amd.engine import Engine, MODE_INT8
    if dims_name == "tiny":
        d, n_samples, B = spec.TINY, 80000, 5
    else:
        d, n_samples, B = replace(spec.FULL, enc_layers=1, dec_layers=2, vocab=1024, audio_token_id=1000, eos_ids=()), 160000, 5
    e = Engine(d, 0, MODE_INT8, max_batch=8, max_ctx=320)
    e.load_synthetic(11)
    segs = [synth.synth_pcm(300 + i, n_samples) for i in range(B)]
    prompt = _prompt(n_samples, d)
    n_new = 6
    force = np.random.default_rng(3).integers(2, min(900, d.vocab - 1), (B, n_new)).astype(np.int32)
    e.set_forced_ids(force)
    try:
        ids_a, lg_a = e.transcribe_batch(segs, [prompt] * B, [n_new] * B, want_logits=True)
        e.set_option("i8_no_xq", 1)
        ids_b, lg_b = e.transcribe_batch(segs, [prompt] * B, [n_new] * B, want_logits=True)
        e.set_option("i8_no_xq", 0)
        ids_c, lg_c = e.transcribe_batch(segs, [prompt] * B, [n_new] * B, want_logits=True)    # and the gathered absmax is clean again
    finally:
        e.set_forced_ids(None)
        e.close()
    assert np.isfinite(lg_a).all()
    assert np.array_equal(lg_a, lg_b), float(np.abs(lg_a - lg_b).max())
    assert np.array_equal(lg_a, lg_c)


def test_int8_on_the_fly_quantisation_with_outliers_vs_oracle(orc):
    """The same path when o_proj's input rows DO hold outliers (the synthetic weights give none: attention outputs stay below 3.6): the
    decoder's v_proj weights are scaled so that attention outputs pass 6.0; the consumer's own scan of the row must list them in the order
    the separate pass does (bit-identical logits) and the result must stay on the int8 oracle."""
    from sonicscribe_amd.engine import Engine, MODE_INT8
    d = spec.TINY
    st = synth.synth_state_dict(d, 23, 2)
    dec_v = [k for k in st if k.endswith("self_attn.v_proj.weight") and st[k].shape == (d.dec_kv_heads * d.dec_head_dim, d.dec_d)]
    assert dec_v
    for k in dec_v:
        st[k] = f16(st[k] * 24.0)
    e = Engine(d, 0, MODE_INT8, max_batch=4, max_ctx=320)
    e.load_state_dict(st)
    om = orc.Model(d, st, mode=orc.MODE_INT8)
    segs = [synth.synth_pcm(70 + i, 80000) for i in range(3)]
    prompt = _prompt(80000, d)
    n_new = 5
    force = np.random.default_rng(4).integers(2, 900, (3, n_new)).astype(np.int32)
    e.set_forced_ids(force)
    try:
        _, lg_a = e.transcribe_batch(segs, [prompt] * 3, [n_new] * 3, want_logits=True)
        att = e.debug_read("satt", (3, d.dec_heads * d.dec_head_dim))
        e.set_option("i8_no_xq", 1)
        _, lg_b = e.transcribe_batch(segs, [prompt] * 3, [n_new] * 3, want_logits=True)
    finally:
        e.set_forced_ids(None)
        e.close()
    n_out = int((np.abs(att) >= 6.0).sum())
    print(f"attention output rows of the last step hold {n_out} elements >= 6.0 (absmax {np.abs(att).max():.1f})")
    assert n_out > 0, "the test no longer produces outliers in o_proj's input"
    assert np.array_equal(lg_a, lg_b), float(np.abs(lg_a - lg_b).max())
    worst = 0.0
    for i in range(3):
        feats, mask = orc.logmel(segs[i])
        r = om.transcribe(feats, int(mask.sum()), prompt, n_new, force_ids=force[i])
        worst = max(worst, float(np.abs(lg_a[:, i] - r["step_logits"]).max()))
    print(f"int8 tiny with attention outliers: max|dlogit| vs oracle {worst:.4f}")
    assert worst <= 0.2


@pytest.mark.parametrize("ln_scale", [1.0, 3.5])
def test_int8_layernorm_quantises_its_rows(orc, ln_scale):
    """int8 encoder / prefill: the LayerNorm in front of q/k/v and fc1 (the RMSNorm in front of q/k/v and gate/up) writes the row's absmax,
    int8 codes and outlier flags in the pass that writes the row; only the list + fix-up passes remain (`i8_no_lnq=1`: the three streaming
    passes of round 2).  Bit-identical logits, also when
    LayerNorm outputs pass 6.0 (weights scaled by 3.5: rows of a request flag columns for each other, which the fix-up pass must zero in
    the rows that had quantised them), a two-window request included; and the result stays on the int8 oracle."""
    from sonicscribe_amd.engine import Engine, MODE_INT8
    d = spec.TINY
    st = synth.synth_state_dict(d, 29, 2)
    n_scaled = 0
    for k in st:
        if ".layers." in k and k.endswith("layernorm.weight"):          # encoder LayerNorms and decoder RMSNorms in front of linears
            st[k] = f16(st[k] * (ln_scale if "audio_tower" in k else min(ln_scale, 2.0))); n_scaled += 1      # (RMSNorm x 2: ~0.3 % of its outputs pass 6.0)
    assert n_scaled == 2 * d.enc_layers + 2 * d.dec_layers
    e = Engine(d, 0, MODE_INT8, max_batch=4, max_ctx=1024)
    e.load_state_dict(st)
    om = orc.Model(d, st, mode=orc.MODE_INT8)
    from sonicscribe_amd import frontend
    short = synth.synth_pcm(80, 80000)
    n_long = 40 * 16000
    long_pcm = frontend.normalise_to_int16(synth.synth_pcm(81, n_long).astype(np.float32) / 32768.0)
    wins = [long_pcm[s0:e0] for s0, e0 in frontend.split_windows(n_long, d)]
    assert len(wins) == 2
    n_audio, _ = frontend.request_audio_tokens(n_long, d)
    p_s = _prompt(80000, d)
    p_l = [1, 17, 23, 5] + [d.audio_token_id] * n_audio + [7, 301, 302, 303, 9, 11]
    n_new = 4
    force = np.random.default_rng(6).integers(2, 900, (2, n_new)).astype(np.int32)
    e.set_forced_ids(force)
    try:
        _, lg_a = e.transcribe_batch([short] + wins, [p_s, p_l], [n_new, n_new], req_win=[0, 1, 3], want_logits=True)
        e.set_option("i8_no_lnq", 1)
        _, lg_b = e.transcribe_batch([short] + wins, [p_s, p_l], [n_new, n_new], req_win=[0, 1, 3], want_logits=True)
    finally:
        e.set_forced_ids(None)
        e.close()
    assert np.array_equal(lg_a, lg_b), float(np.abs(lg_a - lg_b).max())
    f0, m0 = orc.logmel(short)
    r = om.transcribe(f0, int(m0.sum()), p_s, n_new, force_ids=force[0])
    worst = float(np.abs(lg_a[:, 0] - r["step_logits"]).max())
    print(f"int8 tiny, LayerNorm weights x {ln_scale}: max|dlogit| vs oracle {worst:.4f}")
    assert worst <= 0.25


@pytest.mark.parametrize("ln_scale", [1.0, 3.5])
def test_int8_fused_qkv_rope_vt_equals_separate_passes(orc, ln_scale):
    """int8 encoder at full width (d = 1280, 20 heads): the fused q|k|v Linear8bitLt runs on the 256x256 kernel with the partial RoPE applied on
    the way out of the staged tile and V transposed while staging (`i8_no_qkv_fuse=1`: row-major Q|K|V + a RoPE pass + a transpose pass).  Same
    arithmetic in both: bit-identical logits, also when the LayerNorm output holds outlier columns (weights x 3.5); and on the int8 oracle."""
    from sonicscribe_amd.engine import Engine, MODE_INT8
    d = replace(spec.FULL, enc_layers=1, dec_layers=1, vocab=1024, audio_token_id=1000, eos_ids=(990, 991, 992))
    st = {}
    for name, shape, kind in spec.tensor_inventory(d):
        scale, offset = synth.kind_params(kind, shape)
        st[name] = orc.synth_fill(31, name, int(np.prod(shape)), scale, offset, 2).reshape(shape)
    for k in st:
        if "audio_tower.layers" in k and k.endswith("input_layernorm.weight"):
            st[k] = f16(st[k] * ln_scale)
    e = Engine(d, 0, MODE_INT8, max_batch=2, max_ctx=320)
    e.load_state_dict(st)
    segs = [synth.synth_pcm(90 + i, 320000 if i == 0 else 100000) for i in range(2)]
    prompts = [_prompt(len(x), d) for x in segs]
    n_new = 2
    force = np.asarray([[100, 200], [400, 500]], np.int32)
    e.set_forced_ids(force)
    try:
        _, lg_a = e.transcribe_batch(segs, prompts, [n_new, n_new], want_logits=True)
        e.set_option("i8_no_qkv_fuse", 1)
        _, lg_b = e.transcribe_batch(segs, prompts, [n_new, n_new], want_logits=True)
    finally:
        e.set_forced_ids(None)
        e.close()
    assert np.isfinite(lg_a).all()
    assert np.array_equal(lg_a, lg_b), float(np.abs(lg_a - lg_b).max())
    om = orc.Model(d, st, mode=orc.MODE_INT8)
    feats, mask = orc.logmel(segs[1])
    r = om.transcribe(feats, int(mask.sum()), prompts[1], n_new, force_ids=force[1])
    worst = float(np.abs(lg_a[:, 1] - r["step_logits"]).max())
    print(f"int8 full-width 1+1, LayerNorm x {ln_scale}: max|dlogit| vs oracle {worst:.4f}")
    assert worst <= 0.25


def test_int8_bench_config_full_depth_vs_oracle(orc):
    """BASELINE config 4 at its real size: 32 + 28 layers, vocabulary 59264, INT8 mode, 64 x 20 s segments in one batch (what
    `bench.py --mode int8 --batch 64` times).  Two steps under teacher forcing:
      * rows 0 and 41: prefill + decode logits against the int8 oracle at full depth;
      * rows 0, 21, 42, 63: the batch result against single-segment runs (outlier columns are found per request, so a request's
        result must not depend on its neighbours).
    Bound: the fp16 ops between the linears differ in summation order from the oracle's, and one flipped fp16 ulp can move an int8
    code; over 60 layers the logits (|x| <= ~5) are held to 0.25; the measured value is printed."""
    from sonicscribe_amd.engine import Engine, MODE_INT8
    d = spec.FULL
    B, n_new, n_samples = 64, 2, 320000
    e = Engine(d, 0, MODE_INT8, max_batch=B, max_ctx=512)
    e.load_synthetic(SEED)
    assert e.weight_bytes() < 3.7 * 2 ** 30, e.weight_bytes() / 2 ** 30      # round 5: the decoder's row-major int8 matrices are gone (3.60 GiB; 4.85 before)
    segs = [synth.synth_pcm(i, n_samples) for i in range(B)]
    prompt = _prompt(n_samples, d)
    rng = np.random.default_rng(77)
    bad = set(d.eos_ids) | {d.audio_token_id}
    force = np.asarray([[t for t in rng.integers(2, d.vocab, 8) if int(t) not in bad][:n_new] for _ in range(B)], np.int32)
    e.set_forced_ids(force)
    ids, logits = e.transcribe_batch(segs, [prompt] * B, [n_new] * B, want_logits=True)
    assert all(np.array_equal(ids[r], force[r]) for r in range(B)) and np.isfinite(logits).all()
    worst_single = 0.0
    for r in (0, 21, 42, 63):
        e.set_forced_ids(force[r:r + 1])
        _, l1 = e.transcribe_batch([segs[r]], [prompt], [n_new], want_logits=True)
        worst_single = max(worst_single, float(np.abs(l1[:, 0] - logits[:, r]).max()))
    e.set_forced_ids(None)
    e.close()
    st = {}
    for name, shape, kind in spec.tensor_inventory(d):
        scale, offset = synth.kind_params(kind, shape)
        st[name] = orc.synth_fill(SEED, name, int(np.prod(shape)), scale, offset, 2).reshape(shape)
    om = orc.Model(d, st, mode=orc.MODE_INT8)
    del st
    worst = 0.0
    for r in (0, 41):
        feats, mask = orc.logmel(segs[r])
        ref = om.transcribe(feats, int(mask.sum()), prompt, n_new, force_ids=force[r])
        worst = max(worst, float(np.abs(logits[:, r] - ref["step_logits"]).max()))
    print(f"int8 full depth, batch 64: max|dlogit| vs oracle {worst:.4f}, batch vs single {worst_single:.4f}, "
          f"logit range [{logits.min():.2f}, {logits.max():.2f}]")
    assert worst <= 0.25, worst
    assert worst_single == 0.0, worst_single       # outlier columns and scales are per request: a request's bits do not depend on its batch


def test_asrmodel_int8_mode():
    from sonicscribe_amd.asr import ASRModel
    m = ASRModel.from_synthetic(spec.TINY, mode="int8", max_batch=4, max_ctx=512)
    wav = synth.synth_pcm(70, 80000).astype(np.float32) / 32768.0
    t = m.transcribe(wav[None], 16000, max_new_tokens=6)
    assert isinstance(t, str) and len(t.split()) >= 1
    info = m.get_model_info()
    assert info["mode"] == "int8" and info["model_dtype"] == "torch.float16"
    m.close()
