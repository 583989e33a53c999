"""The 1 - 4 row token step as weight-streaming GEMV kernels (csrc/gemv.hip; engine option decode_gemv; round 6) - a MEASURED AND LOST experiment, kept
selectable like the others (profiles/round6_gemv_ab.txt): five launches per decoder layer, no MFMA tile machinery, every projection sees its whole K in one block - and 6 % more
time per layer at one row than the MFMA chain (38.6 against 36.3 us: gate/up 12.4 against 10.7, down_proj 8.2 against 6.4), because at one row the MFMA kernels are already bound by
the weight stream (gate/up: 50 MB in 10.7 us = 4.7 TB/s), not by their tile machinery.  The chain keeps the reference's rounding points (modeling_llama.py:53-67, 163-176,
217-324) and differs from the default chain only in the order an output's K products are added.  What this file holds it to:
  * against the CPU oracle (bf16 / fp16 modes): logits within the bound the default chain is held to, token ids equal outside near-ties;
  * against the default chain: within a few ulp of the element type, NOT bit-identical (it is another summation order - and so it is known to have run);
  * deterministic, and a row's bits do not depend on the other rows of its step (1 .. 4 rows);
  * through continuous loops (R = 2 / R = 4 chunk graphs): the same tokens as the batch entry point in the same mode."""
from dataclasses import replace

import numpy as np
import pytest

from sonicscribe_amd import spec, synth

pytestmark = pytest.mark.gpu
SEED = 20260128


def prompt_for(d, n, tail=()):
    return [1, 17, 23, 5] + [d.audio_token_id] * spec.audio_token_count(spec.valid_frames(n)) + [7, 301, 302, 303, 9, 11] + list(tail)


FULLW = replace(spec.FULL, enc_layers=1, dec_layers=2, vocab=1024, audio_token_id=1000, eos_ids=())     # full-width layers (the shapes the GEMV chain is built for), shallow


@pytest.mark.parametrize("tag", ["fullwidth-bf16", "fullwidth-fp16"])
def test_gemv_chain_against_oracle_and_default_chain(tag):
    from oracle import oracle
    from sonicscribe_amd.engine import Engine, MODE_F16, MODE_NATIVE
    fp16 = tag.endswith("fp16")
    d = FULLW
    e = Engine(d, 0, MODE_F16 if fp16 else MODE_NATIVE, max_batch=8, max_ctx=512)
    e.load_synthetic(SEED)
    segs = [synth.synth_pcm(30 + i, 16000 * (2 + 3 * i)) for i in range(4)]
    prompts = [prompt_for(d, len(s), tail=[40 + i] * i) for i, s in enumerate(segs)]
    n_new = 10
    rng = np.random.default_rng(8)
    force = rng.integers(2, min(d.vocab, 900), size=(4, n_new)).astype(np.int32)
    force[force == d.audio_token_id] = 7

    def run(rows, gemv):
        e.set_option("decode_gemv", int(gemv))
        e.set_forced_ids(force[rows])
        try:
            ids, lg = e.transcribe_batch([segs[r] for r in rows], [prompts[r] for r in rows], [n_new] * len(rows), want_logits=True)
            per_layer = e.timings()["decode_launches_per_layer"]
        finally:
            e.set_forced_ids(None)
            e.set_option("decode_gemv", 0)
        return lg, per_layer
    lg_d, _ = run([0, 1, 2, 3], False)
    lg_g, per_layer = run([0, 1, 2, 3], True)
    assert per_layer == 5
    ulp = 2.0 ** -9 if fp16 else 2.0 ** -6                     # one step of the element type at |logit| in [2, 4)
    dd = float(np.abs(lg_g - lg_d).max())
    print(f"{tag}: GEMV chain vs default chain over {n_new} teacher-forced steps x 4 rows: max |dlogit| {dd:.5f} ({dd / ulp:.1f} ulp); logits in [{lg_d.min():.2f}, {lg_d.max():.2f}]")
    assert np.array_equal(lg_g[0], lg_d[0])                    # step 0 is the prefill: the same kernels
    assert 0 < dd <= 6 * ulp                                   # another summation order: close, and not the same bits
    sd = synth.synth_state_dict(d, SEED, bf16=2 if fp16 else True)
    om = oracle.Model(d, sd, mode=oracle.MODE_FP16) if fp16 else oracle.Model(d, sd, bf16=True)
    tol = (4 * 2.0 ** -9 * 2) if fp16 else 4 * 2.0 ** -6
    for r in (0, 3):
        feats, mask = oracle.logmel(segs[r])
        o = om.transcribe(feats, int(mask.sum()), prompts[r], n_new, force_ids=force[r])
        err = float(np.abs(lg_g[:, r] - o["step_logits"]).max())
        print(f"  row {r} vs the oracle: max |dlogit| {err:.5f} (bound {tol:.5f})")
        assert err <= tol
    # a row's bits do not depend on its neighbours, and the chain is deterministic
    for rows in ([0], [2], [1, 3], [0, 1, 2]):
        lg_s, _ = run(rows, True)
        for k, r in enumerate(rows):
            assert np.array_equal(lg_s[:, k].view(np.uint32), lg_g[:, r].view(np.uint32)), (rows, r)
    # free running: ids equal the default chain's wherever the default chain's margin is not a near-tie
    e.set_option("decode_gemv", 1)
    ids_g, lgf = e.transcribe_batch(segs[:2], prompts[:2], [12, 12], want_logits=True)
    e.set_option("decode_gemv", 0)
    ids_d, lgd = e.transcribe_batch(segs[:2], prompts[:2], [12, 12], want_logits=True)
    for r in range(2):
        srt = np.sort(lgd[:, r], axis=1)
        safe = (srt[:, -1] - srt[:, -2]) > 8 * ulp
        n_safe = len(safe) if safe.all() else int(np.argmin(safe))
        assert np.array_equal(ids_g[r][:n_safe], ids_d[r][:n_safe]), (r, ids_g[r], ids_d[r])
    e.close()


def test_gemv_chain_in_continuous_loops():
    from sonicscribe_amd.engine import Engine
    d = FULLW
    e = Engine(d, 0, max_batch=8, max_ctx=512)
    e.load_synthetic(SEED)
    e.set_option("decode_gemv", 1)
    segs = [synth.synth_pcm(60 + i, 16000 * (2 + i)) for i in range(4)]
    prompts = [prompt_for(d, len(s)) for s in segs]
    solo = [e.transcribe_batch([s], [p], [14])[0][0] for s, p in zip(segs, prompts)]
    pre = e.slot()
    e.service_begin()
    got, seq = {}, {}
    for step in range(300):
        if step < 4:                                            # rows join one by one: R = 2 graphs, then R = 4
            pre.stage_pcm(segs[step:step + 1]); pre.prefill(prompts[step:step + 1], [14]); seq[step] = e.splice_rows(pre, [0], [step])
        live = [r for r in seq if r not in got]
        if not live and step >= 4:
            break
        fin, nn, s_, _ = e.service_step(1, max(live) + 1 if live else 1)
        for r in live:
            if s_ > seq[r] and fin[r]:
                got[r] = e.fetch_row(r, int(nn[r]))
    e.service_end()
    for r in range(4):
        assert np.array_equal(got[r], solo[r]), r
    e.close()
