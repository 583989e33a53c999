"""The production kernel templates at fp16's rounding (SONIC_MODE_F16, test only; include/sonic_hip.h).

north_star asks for logits within 1e-3 of the reference; the bf16 path is within 4-8 bf16 ulp (0.06-0.12) of the bf16 reference arithmetic,
which says nothing about errors smaller than that.  The same templates compiled for IEEE half (their KF16 / f16_t instantiations: 128x128 and
256x256 GEMMs with every epilogue, flash and decode attention, norms, RoPE + KV append, the skinny decode GEMMs, the greedy controller and -
since round 5 - the fused decode kernels skinny_o / skinny_gu that the bf16 headline runs: VERDICT r4 item 7) are checked here
  * against the oracle in fp16 mode - same rounding points, only the summation order differs: a few fp16 ulp;
  * against the oracle in FP32 mode with the same weights (bf16-valued, exact in fp16) = the reference's fp32 arithmetic (the oracle's fp32
    mode is pinned to transformers' fp32 generate() at 1e-3, tests/test_oracle_golden.py): what is left is fp16 activation rounding.
A layout, mask, position, RoPE-pairing, KV-append or argmax bug shows at O(0.1-1) and at 1e-2 alike; this bound is 8-16 x tighter than bf16's.
The fp32 kind (1e-3 literally) is tests/test_gpu_fp32_mode.py (SONIC_MODE_F32, round 6): its own plain-fp32 kernels behind the same plan / staging /
controller, because every product kernel's LDS image assumes 2-byte elements; THIS file is what checks the product templates themselves below bf16."""
from dataclasses import replace

import numpy as np
import pytest

from sonicscribe_amd import frontend, spec, synth

pytestmark = pytest.mark.gpu
SEED = 20260128


def f16_ulp(x):
    x = np.maximum(np.abs(np.asarray(x, np.float64)), 2.0 ** -14)
    return 2.0 ** (np.floor(np.log2(x)) - 10)


def prompt_for(d, n):
    return [1, 17, 23, 5] + [d.audio_token_id] * spec.audio_token_count(spec.valid_frames(n)) + [7, 301, 302, 303, 9, 11]


def check(tag, hip, o16, o32, ulps16, abs32):
    d16 = np.abs(hip - o16)
    worst16 = float(d16.max() / f16_ulp(np.abs(o32).max()))          # in fp16 ulps of the largest logit (the bf16 tests' convention)
    d32 = float(np.abs(hip - o32).max())
    ref = float(np.abs(o16 - o32).max())
    print(f"{tag}: max |HIP fp16 - oracle fp16| {d16.max():.5f} ({worst16:.1f} fp16 ulp); max |HIP fp16 - fp32 arithmetic| {d32:.5f} "
          f"(oracle fp16 vs fp32: {ref:.5f}); logits in [{o32.min():.2f}, {o32.max():.2f}]")
    assert worst16 <= ulps16, (tag, worst16)
    assert d32 <= abs32, (tag, d32)


def test_tiny_prefill_forced_steps_and_two_windows():
    from oracle import oracle
    from sonicscribe_amd.engine import Engine, MODE_F16
    d = replace(spec.TINY, eos_ids=())
    e = Engine(d, 0, MODE_F16, max_batch=4, max_ctx=1024)
    e.load_synthetic(SEED)
    sd = synth.synth_state_dict(d, SEED, bf16=2)                 # bf16 values held as fp16: exact, the engine's weights
    o16 = oracle.Model(d, sd, mode=oracle.MODE_FP16)
    o32 = oracle.Model(d, sd, mode=oracle.MODE_FP32)
    rng = np.random.default_rng(3)
    segs = [synth.synth_pcm(10, 80000), synth.synth_pcm(11, 320000), synth.synth_pcm(12, 20480)]
    prompts = [prompt_for(d, len(s)) for s in segs]
    n_new = 24
    force = rng.integers(2, d.vocab, size=(3, n_new)).astype(np.int32)
    force[force == d.audio_token_id] = 7
    e.set_forced_ids(force)
    ids, logits = e.transcribe_batch(segs, prompts, [n_new] * 3, want_logits=True)
    e.set_forced_ids(None)
    for r in range(3):
        feats, mask = oracle.logmel(segs[r])
        a = o16.transcribe(feats, int(mask.sum()), prompts[r], n_new, force_ids=force[r])
        b = o32.transcribe(feats, int(mask.sum()), prompts[r], n_new, force_ids=force[r])
        assert np.array_equal(ids[r], force[r])
        check(f"tiny row {r}, prefill + {n_new - 1} teacher-forced steps", logits[:, r], a["step_logits"], b["step_logits"], ulps16=4.0, abs32=8e-3)
    # free running: ids equal the fp16 oracle's wherever its top-1 / top-2 margin is not a near-tie
    ids_f, lg_f = e.transcribe_batch(segs[:2], prompts[:2], [12, 12], want_logits=True)
    for r in range(2):
        feats, mask = oracle.logmel(segs[r])
        a = o16.transcribe(feats, int(mask.sum()), prompts[r], 12)
        srt = np.sort(a["step_logits"], axis=1)
        safe = (srt[:, -1] - srt[:, -2]) > 8 * f16_ulp(srt[:, -1])
        n_safe = len(safe) if safe.all() else int(np.argmin(safe))
        assert np.array_equal(ids_f[r][:n_safe], a["new_ids"][:n_safe]), (r, ids_f[r], a["new_ids"])
    # one request of two windows (35 s)
    pcm = synth.synth_pcm(40, 560000)
    wins = [pcm[s:t] for s, t in frontend.split_windows(len(pcm), d)]
    total, per_win = frontend.request_audio_tokens(len(pcm), d)
    prompt = [1, 17] + [d.audio_token_id] * total + [7, 9]
    f2 = rng.integers(2, 900, size=(1, 10)).astype(np.int32)
    e.set_forced_ids(f2)
    _, lg = e.transcribe_batch(wins, [prompt], [10], req_win=[0, 2], want_logits=True)
    e.set_forced_ids(None)
    fm = [oracle.logmel(w) for w in wins]
    feats = np.stack([f for f, _ in fm]); nv = [int(m.sum()) for _, m in fm]
    a = o16.transcribe(feats, nv, prompt, 10, force_ids=f2[0]); b = o32.transcribe(feats, nv, prompt, 10, force_ids=f2[0])
    check("tiny two-window request, 10 forced steps", lg[:, 0], a["step_logits"], b["step_logits"], ulps16=4.0, abs32=8e-3)
    e.close()


def test_full_width_layers_vocab_59264():
    """full-width layers (1280 / 5120 / 20 heads, 2048 / 6144 GQA 16:4, vocabulary 59264) at depth 1 + 1: the 256x256 GEMMs, the head-dim-64
    flash kernel at T = 1500 and the 59264-wide lm_head in their fp16 instantiations"""
    from oracle import oracle
    from sonicscribe_amd.engine import Engine, MODE_F16
    d = replace(spec.FULL, enc_layers=1, dec_layers=1, eos_ids=())
    e = Engine(d, 0, MODE_F16, max_batch=2, max_ctx=512)
    e.load_synthetic(SEED)
    sd = synth.synth_state_dict(d, SEED, bf16=2)
    o16 = oracle.Model(d, sd, mode=oracle.MODE_FP16)
    o32 = oracle.Model(d, sd, mode=oracle.MODE_FP32)
    seg = synth.synth_pcm(21, 5 * 16000)
    prompt = prompt_for(d, len(seg))
    force = np.asarray([[40, 4100, 59000, 77, 12345, 31000]], np.int32)
    e.set_forced_ids(force)
    _, lg = e.transcribe_batch([seg], [prompt], [6], want_logits=True)
    e.set_forced_ids(None)
    feats, mask = oracle.logmel(seg)
    a = o16.transcribe(feats, int(mask.sum()), prompt, 6, force_ids=force[0]); b = o32.transcribe(feats, int(mask.sum()), prompt, 6, force_ids=force[0])
    check("full-width 1 + 1 layers, 6 forced steps", lg[:, 0], a["step_logits"], b["step_logits"], ulps16=4.0, abs32=8e-3)
    # the decode steps above went through the FUSED decode kernels (skinny_o + skinny_gu with RMSNorm in LDS: eligible at this width, and since
    # round 5 instantiated for fp16 as well - they are what the bf16 headline runs).  The unfused forms must meet the same bounds, and the two
    # paths must not be the same computation (different reduction trees: some logits differ in their last bits)
    e.set_option("no_fused_gu", 1)
    e.set_forced_ids(force)
    _, lg_u = e.transcribe_batch([seg], [prompt], [6], want_logits=True)
    e.set_forced_ids(None)
    e.set_option("no_fused_gu", 0)
    check("full-width 1 + 1 layers, unfused decode path", lg_u[:, 0], a["step_logits"], b["step_logits"], ulps16=4.0, abs32=8e-3)
    assert np.array_equal(lg_u[0], lg[0])                              # (step 0 is the prefill: no decode kernel involved)
    assert not np.array_equal(lg_u[1:], lg[1:]), "the fused decode kernels were not taken in fp16 mode"
    assert np.abs(lg_u - lg).max() <= 8 * f16_ulp(np.abs(b["step_logits"]).max())
    e.close()
