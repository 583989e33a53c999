"""GPU parity at the sizes bench.py times (BASELINE config 2: 32 x 20 s, GLM-ASR-Nano at full depth and full vocabulary), plus the
controller cases the reference's greedy loop defines (first-maximum argmax, EOS stop, ragged budgets, multi-window requests) --
all against the CPU oracle / the reference fixtures, through the C ABI.

Tolerances: bf16 kernels reproduce torch's op-boundary rounding; what remains is fp32 accumulation order (isolated 1-ulp bf16
flips that compound with depth).  One bf16 ulp is 2^-6 for |logit| in [2, 4) and 2^-5 in [4, 8).
"""
import os
from dataclasses import replace

import numpy as np
import pytest

from sonicscribe_amd import spec, synth

pytestmark = pytest.mark.gpu

SEED = 20260128


def bf(x):
    return synth.round_bf16(np.asarray(x, np.float32))


@pytest.fixture(scope="module")
def eng():
    from sonicscribe_amd.engine import Engine
    e = Engine(spec.TINY, 0, max_batch=16, max_ctx=512)
    e.load_synthetic(SEED)
    yield e
    e.close()


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def tiny_prompt(n_samples, d=spec.TINY, pre=(1, 17, 23, 5), suf=(7, 301, 302, 303, 9, 11)):
    return list(pre) + [d.audio_token_id] * spec.audio_token_count(spec.valid_frames(n_samples)) + list(suf)


# ------------------------------------------------------------------------------------------ lm_head instance + argmax
def test_skinny_lm_head_shape(eng):
    """The exact decode-step lm_head instance of the bench: M = 32 rows, N = 59264 (vocabulary), K = 2048, two K slabs."""
    rng = np.random.default_rng(59264)
    M, N, K = 32, 59264, 2048
    X = bf(rng.standard_normal((M, K))); W = bf(rng.standard_normal((N, K)) * 0.05)
    got = eng.test_skinny(X, W)
    ref = (X.astype(np.float64) @ W.T.astype(np.float64)).astype(np.float32)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-3 * np.sqrt(K) * 0.05)
    # a column per 16-row tile boundary / block edge must not be permuted: spot rows of W that are unique
    assert np.array_equal(np.argmax(got, axis=1), np.argmax(ref, axis=1))


def _argmax_ref(slabs, B):
    s = slabs[0].astype(np.float32).copy()
    for k in range(1, slabs.shape[0]):
        s = (s + slabs[k]).astype(np.float32)                       # fixed order ks = 0, 1, ...
    r = bf(s[:B])
    return np.argmax(r, axis=1), r                                   # numpy argmax = first maximum = torch.argmax


@pytest.mark.parametrize("ks,V", [(2, 59264), (1, 59264), (3, 1024), (2, 4096 + 64)])
def test_greedy_first_max_and_ties(eng, ks, V):
    """greedy_kernel at the bench's vocabulary: first-maximum argmax (torch semantics, generation/utils.py:2925) over the bf16-rounded
    sum of the lm_head's K slabs, with engineered EXACT ties: across slabs, across threads, waves, strides of one thread, and ties
    that only appear after the bf16 rounding."""
    rng = np.random.default_rng(ks * 1000 + V)
    B, mpad = 32, 32
    slabs = (rng.standard_normal((ks, mpad, V)) * 0.3).astype(np.float32)
    big = 7.0

    def put(row, col, total):          # split `total` over the slabs in a column-specific way; the fp32 slab sum is exactly `total`
        parts = np.zeros(ks, np.float32)
        if ks > 1:
            parts[0] = np.float32(0.25 * ((col % 5) + 1)); parts[ks - 1] += np.float32(total) - parts[0]
        else:
            parts[0] = total
        slabs[:, row, col] = parts
    cases = {
        0: [(5, big), (300, big)],                                  # two waves of the first stride
        1: [(300, big), (5, big)],
        2: [(8, big), (9, big)],                                    # neighbours inside one thread's float4
        3: [(V - 1, big), (V - 4, big)],                            # last vector of the row
        4: [(13, big), (13 + 4096, big)] if V > 8192 else [(13, big), (14, big)],          # same thread, two strides of one trip
        5: [(21, big), (21 + 4 * 4096, big)] if V > 5 * 4096 else [(21, big), (22, big)],   # same thread, next trip of the loop
        6: [(1000 % V, big + 0.001), (40, big + 0.002)],            # distinct fp32 sums that round to the same bf16: still a tie
        7: [(V // 2 + 1, big), (V // 2 - 3, big), (3, big - 0.5)],
        8: [(4095, big), (4096, big)] if V > 4100 else [(V // 4 - 1, big), (V // 4, big)],   # last thread of a stride / first of the next
        9: [(63 * 4 + 3, big), (64 * 4, big)],                      # wave boundary
    }
    for row, lst in cases.items():
        for col, val in lst:
            put(row, col, val)
    tok, lg = eng.test_greedy(slabs, B, want_logits=True)
    ref_tok, ref_lg = _argmax_ref(slabs, B)
    assert np.array_equal(lg, ref_lg)                                # the compared logits are bit-exact (fixed summation order)
    assert np.array_equal(tok, ref_tok), (tok, ref_tok)
    for row, lst in cases.items():
        tied = [c for c, v in lst if bf(np.float32(v)) == bf(np.float32(max(v for _, v in lst)))]
        assert tok[row] == min(tied), (row, tok[row], lst)


# ------------------------------------------------------------------------------------------ full depth, full vocabulary, B = 32 x 20 s
def _full_state(orc, d, seed):
    out = {}
    for name, shape, kind in spec.tensor_inventory(d):
        scale, offset = synth.kind_params(kind, shape)
        out[name] = orc.synth_fill(seed, name, int(np.prod(shape)), scale, offset, True).reshape(shape)
    return out


def test_bench_config_full_depth_vs_oracle(orc):
    """BASELINE config 2 at its real size: 32 encoder + 28 decoder layers, vocabulary 59264, 32 x 20 s segments in one batch -- the kernel
    instances bench.py times (gemm256 at M = 48000, flash attention over 32 x 20 heads, prefill at M = 8320, the decode step's fused
    kernels at 32 rows, lm_head + greedy at V = 59264).  Three steps under teacher forcing with a varying id sequence:
      * rows 0 and 17: prefill + decode logits against the bf16 oracle at full depth;
      * every row: the batch result against a single-segment run of the same engine: per-segment results must not depend on batching,
        bit for bit.
    Bound: 8 bf16 ulp of 2^-6 (0.125) on logits of magnitude <= ~5 after 60 layers; measured values are printed."""
    from sonicscribe_amd.engine import Engine
    d = spec.FULL
    B, n_new, n_samples = 32, 3, 320000
    e = Engine(d, 0, max_batch=B, max_ctx=512)
    e.load_synthetic(SEED)
    # one copy of every decoder projection since round 5 (the prefill GEMMs read the decode step's fragment-tiled weights): the bf16 model's 2.25 G
    # parameters + the tiled lm_head copy of the tied embedding + fp32 biases / norm weights - 4.21 GiB, where rounds 1-4 held 8.03
    assert e.weight_bytes() < 4.35 * 2 ** 30, e.weight_bytes() / 2 ** 30
    segs = [synth.synth_pcm(i, n_samples) for i in range(B)]
    prompt = tiny_prompt(n_samples, d)
    rng = np.random.default_rng(31)
    bad = set(d.eos_ids) | {d.audio_token_id}
    force = np.asarray([[t for t in rng.integers(2, d.vocab, 8) if int(t) not in bad][:n_new] for _ in range(B)], np.int32)
    e.set_forced_ids(force)
    ids, logits = e.transcribe_batch(segs, [prompt] * B, [n_new] * B, want_logits=True)
    assert all(np.array_equal(ids[r], force[r]) for r in range(B))
    assert np.isfinite(logits).all()
    tol = 8 * 2.0 ** -6
    # (a) the batch against single-segment runs
    worst_single = 0.0
    for r in range(B):
        e.set_forced_ids(force[r:r + 1])
        _, l1 = e.transcribe_batch([segs[r]], [prompt], [n_new], want_logits=True)
        worst_single = max(worst_single, float(np.abs(l1[:, 0] - logits[:, r]).max()))
    e.set_forced_ids(None)
    # (b) the graph-replayed free-running loop still emits the budget for every row and passes the engine's state invariants
    ids_g, _ = e.transcribe_batch(segs, [prompt] * B, [5] * B)
    assert all(1 <= len(x) <= 5 for x in ids_g)
    e.close()
    # (c) the oracle at full depth for two rows
    om = orc.Model(d, _full_state(orc, d, SEED), bf16=True)
    worst_orc, flips = 0.0, 0
    for r in (0, 17):
        feats, mask = orc.logmel(segs[r])
        ref = om.transcribe(feats, int(mask.sum()), prompt, n_new, force_ids=force[r])
        dl = np.abs(logits[:, r] - ref["step_logits"])
        worst_orc = max(worst_orc, float(dl.max()))
        for s in range(n_new):
            srt = np.sort(ref["step_logits"][s]); margin = srt[-1] - srt[-2]
            if margin > 2 * tol:
                assert int(np.argmax(logits[s, r])) == int(np.argmax(ref["step_logits"][s])), (r, s)   # bit-exact ids outside near-ties
            else:
                flips += 1
    print(f"full depth: max|dlogit| vs oracle {worst_orc:.4f}, batch vs single {worst_single:.4f}, near-tie steps {flips}, "
          f"logit range [{logits.min():.2f}, {logits.max():.2f}]")
    assert worst_orc <= tol, worst_orc
    # by construction, not by tolerance: every kernel sums an output element in an order that does not depend on the batch (fixed slab
    # order, the 128x128 and 256x256 GEMMs accumulate alike and share their epilogue arithmetic), so a segment's logits are the same
    # bits alone or in a batch of 32
    assert worst_single == 0.0, worst_single


# ------------------------------------------------------------------------------------------ trajectories that vary
def test_forced_trajectory_vs_golden_and_oracle(eng, orc, golden_dir):
    """Teacher forcing with a varying id sequence: each step gathers a different embedding row and appends at a new position, so a
    stale-token or off-by-one position bug shows in the next step's logits.  Against the reference fixture (generate() under a
    forcing LogitsProcessor) and the oracle; bound 4 bf16 ulp at |logit| < 4."""
    g = np.load(os.path.join(golden_dir, "tiny_forced_bf16.npz"))
    d = spec.TINY
    om = orc.Model(d, synth.synth_state_dict(d, SEED, bf16=True), bf16=True)
    segs, prompts, forces = [], [], []
    for si in range(2):
        p = f"s{si}_"
        segs.append(synth.synth_pcm(int(g[p + "seg_index"]), int(g[p + "n_samples"])))
        prompts.append(g[p + "prompt_ids"]); forces.append(g[p + "force_ids"])
    force = np.stack(forces)
    n_new = force.shape[1]
    eng.set_forced_ids(force)
    try:
        ids, logits = eng.transcribe_batch(segs, prompts, [n_new, n_new], want_logits=True)
    finally:
        eng.set_forced_ids(None)
    tol = 4 * 2.0 ** -6
    for si in range(2):
        p = f"s{si}_"
        assert np.array_equal(ids[si], force[si])
        assert np.abs(logits[:, si] - g[p + "step_logits"]).max() <= tol
        feats, mask = orc.logmel(segs[si])
        r = om.transcribe(feats, int(mask.sum()), prompts[si], n_new, force_ids=force[si])
        assert np.abs(logits[:, si] - r["step_logits"]).max() <= tol


def test_eos_stop_vs_oracle(orc):
    """EOS handling against the oracle with the SAME EOS set (generation/utils.py:2928-2936): the forced sequence carries an EOS id at
    step 3 of row 0 and none for row 1; row 0 must stop right after emitting it, row 1 runs to its budget, logits agree up to the stop."""
    from sonicscribe_amd.engine import Engine
    d = replace(spec.TINY, eos_ids=(555, 991, 992))
    e = Engine(d, 0, max_batch=2, max_ctx=512)
    e.load_synthetic(SEED)
    om = orc.Model(d, synth.synth_state_dict(d, SEED, bf16=True), bf16=True)
    segs = [synth.synth_pcm(31, 80000), synth.synth_pcm(32, 48000)]
    prompts = [tiny_prompt(len(s), d) for s in segs]
    force = np.asarray([[40, 41, 42, 555, 43, 44, 45, 46], [60, 61, 62, 63, 64, 65, 66, 67]], np.int32)
    e.set_forced_ids(force)
    ids, logits = e.transcribe_batch(segs, prompts, [8, 6], want_logits=True)
    e.set_forced_ids(None)
    tol = 4 * 2.0 ** -6
    for r in range(2):
        feats, mask = orc.logmel(segs[r])
        ref = om.transcribe(feats, int(mask.sum()), prompts[r], [8, 6][r], force_ids=force[r])
        assert np.array_equal(ids[r], ref["new_ids"]), (ids[r], ref["new_ids"])
        assert np.abs(logits[: len(ids[r]), r] - ref["step_logits"]).max() <= tol
    assert ids[0].tolist() == [40, 41, 42, 555] and len(ids[1]) == 6
    # free-running: make the model's own favourite token an EOS and compare the stop step with the oracle
    free, _ = e.transcribe_batch([segs[0]], [prompts[0]], [8])
    d2 = replace(spec.TINY, eos_ids=(int(free[0][2]), 991, 992))
    e2 = Engine(d2, 0, max_batch=2, max_ctx=512)
    e2.load_synthetic(SEED)
    om2 = orc.Model(d2, synth.synth_state_dict(d2, SEED, bf16=True), bf16=True)
    got, _ = e2.transcribe_batch([segs[0]], [prompts[0]], [8])
    feats, mask = orc.logmel(segs[0])
    ref = om2.transcribe(feats, int(mask.sum()), prompts[0], 8)
    assert np.array_equal(got[0], ref["new_ids"]) and len(got[0]) < 8
    e.close(); e2.close()


def test_multi_window_request_vs_golden_and_oracle(eng, orc, golden_dir):
    """35 s of audio = two windows behind one prompt (processing_glmasr.py:136-176): logits against the reference fixture and the oracle."""
    from sonicscribe_amd import frontend
    d = spec.TINY
    g = np.load(os.path.join(golden_dir, "tiny_multi_bf16.npz"))
    pcm = synth.synth_pcm(int(g["seg_index"]), int(g["n_samples"]))
    wins = [pcm[s:e] for s, e in frontend.split_windows(len(pcm), d)]
    force = g["force_ids"][None]
    eng.set_forced_ids(force)
    try:
        ids, logits = eng.transcribe_batch(wins, [g["prompt_ids"]], [force.shape[1]], req_win=[0, 2], want_logits=True)
    finally:
        eng.set_forced_ids(None)
    tol = 4 * 2.0 ** -6
    assert np.abs(logits[:, 0] - g["step_logits"]).max() <= tol
    om = orc.Model(d, synth.synth_state_dict(d, SEED, bf16=True), bf16=True)
    fm = [orc.logmel(w) for w in wins]
    r = om.transcribe(np.stack([f for f, _ in fm]), [int(m.sum()) for _, m in fm], g["prompt_ids"], force.shape[1], force_ids=force[0])
    assert np.abs(logits[:, 0] - r["step_logits"]).max() <= tol
    # and mixed with a single-window request in the same batch
    seg2 = synth.synth_pcm(41, 64000)
    ids2, _ = eng.transcribe_batch(wins + [seg2], [g["prompt_ids"], tiny_prompt(len(seg2))], [4, 4], req_win=[0, 2, 3])
    alone, _ = eng.transcribe_batch([seg2], [tiny_prompt(len(seg2))], [4])
    assert np.array_equal(ids2[1], alone[0])


def test_mixed_budgets_near_max_ctx():
    """[long prompt, small budget] next to [short prompt, large budget] close to max_ctx = 512: before finished rows froze their
    context the long row's position ran to prompt + max(max_new) and left its KV-cache region.  Every row must equal its
    single-request run, and the engine's state invariants (checked in sonic_fetch_tokens) must hold.  (No EOS ids: every row runs
    to its budget.)"""
    from sonicscribe_amd.engine import Engine
    eng = Engine(replace(spec.TINY, eos_ids=()), 0, max_batch=4, max_ctx=512)
    eng.load_synthetic(SEED)
    segs = [synth.synth_pcm(90, 480000), synth.synth_pcm(91, 16000), synth.synth_pcm(92, 48000)]
    prompts = [tiny_prompt(len(s)) for s in segs]
    budgets = [512 - len(prompts[0]), 512 - len(prompts[1]), 40]
    assert len(prompts[0]) + budgets[1] > 512                      # the overrun the old controller produced
    ids, _ = eng.transcribe_batch(segs, prompts, budgets)
    for i in range(3):
        assert len(ids[i]) == budgets[i]
        alone, _ = eng.transcribe_batch([segs[i]], [prompts[i]], [budgets[i]])
        assert np.array_equal(alone[0], ids[i]), i
    # the cache regions of the neighbours are intact: a fresh run of the batch in the opposite order gives the same rows
    ids_r, _ = eng.transcribe_batch(segs[::-1], prompts[::-1], budgets[::-1])
    for i in range(3):
        assert np.array_equal(ids_r[2 - i], ids[i])
    eng.close()


def test_two_engines_in_one_process():
    """Two engines in one process (the in-process multi-GPU dispatcher creates one per device; on a one-GPU box both sit on device 0):
    interleaved calls, per-engine knobs, results equal to each engine's own isolated run."""
    from sonicscribe_amd.engine import Engine
    a = Engine(spec.TINY, 0, max_batch=4, max_ctx=512); a.load_synthetic(SEED)
    d2 = replace(spec.TINY, dec_ff=1024, enc_ff=256)
    b = Engine(d2, 0, max_batch=4, max_ctx=512); b.load_synthetic(7)
    seg = synth.synth_pcm(5, 80000)
    pa, pb = tiny_prompt(len(seg)), tiny_prompt(len(seg), d2)
    ra0, _ = a.transcribe_batch([seg], [pa], [6]); rb0, _ = b.transcribe_batch([seg], [pb], [6])
    b.set_option("no_fused_gu", 1)                                   # must not leak into engine a
    for _ in range(3):
        ra, _ = a.transcribe_batch([seg], [pa], [6]); rb, _ = b.transcribe_batch([seg], [pb], [6])
        assert np.array_equal(ra[0], ra0[0])
    b.set_option("no_fused_gu", 0)
    rb, _ = b.transcribe_batch([seg], [pb], [6])
    assert np.array_equal(rb[0], rb0[0])
    a.close(); b.close()


def test_two_engines_capture_and_fetch_concurrently():
    """Two engines on one GPU driven from two threads, every call with a batch size the engine has not seen yet (so each call captures a
    decode graph) while the other thread is fetching results: the device -> host copies must not touch the legacy stream (the
    synchronous hipMemcpy failed here with 'operation would make the legacy stream depend on a capturing blocking stream').  Results
    equal the sequential ones."""
    import threading
    from sonicscribe_amd.engine import Engine
    engs = [Engine(spec.TINY, 0, max_batch=8, max_ctx=512) for _ in range(2)]
    for e in engs:
        e.load_synthetic(SEED)
    segs = [synth.synth_pcm(90 + i, 16000 * (2 + i % 3)) for i in range(8)]
    prompts = [tiny_prompt(len(s)) for s in segs]
    want = [engs[0].transcribe_batch([segs[i]], [prompts[i]], [6])[0][0] for i in range(8)]
    for e in engs:
        e.set_option("no_graph", 0)                      # (drops the graphs captured above: every batch size below captures again)
    got = [[None] * 8 for _ in engs]
    errs = []

    def work(k):
        try:
            order = range(1, 9) if k == 0 else range(8, 0, -1)
            for n in order:                              # batch sizes 1..8, opposite orders in the two threads
                ids, _ = engs[k].transcribe_batch(segs[:n], prompts[:n], [6] * n)
                for i in range(n):
                    assert np.array_equal(ids[i], want[i]), (k, n, i)
                got[k][n - 1] = True
        except BaseException as ex:
            errs.append(ex)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs
    assert all(all(g) for g in got)
    for e in engs:
        e.close()


def test_multi_replica_dispatch_and_async_entry():
    """Two engine replicas in ONE process behind the dispatcher (the in-process multi-GPU form; on a one-GPU box both replicas sit on
    device 0 via device="cuda:0,0"), fed through submit() / transcribe_async() the way the WebSocket callers would: results equal the
    single-replica results, both replicas work, partials and finals do not share device batches."""
    import asyncio
    from sonicscribe_amd.asr import ASRModel
    one = ASRModel.from_synthetic(spec.TINY, device="cuda:0", max_batch=4, max_ctx=512)
    two = ASRModel.from_synthetic(spec.TINY, device="cuda:0,0", max_batch=4, max_ctx=512)
    assert len(two.models) == 2 and two.get_model_info()["replicas"] == 2
    wavs = [synth.synth_pcm(200 + i, 16000 * (2 + i % 3)).astype(np.float32) / 32768.0 for i in range(12)]
    want_p = [one.transcribe(w[None, -20480:], 16000, max_new_tokens=15) for w in wavs]
    want_f = [one.transcribe(w[None], 16000, max_new_tokens=40) for w in wavs]

    async def run():
        part = [two.transcribe_async(w[None, -20480:], 16000, 15, session=f"c{i}") for i, w in enumerate(wavs)]
        fin = [two.transcribe_async(w[None], 16000, 40, session=f"c{i}") for i, w in enumerate(wavs)]
        return await asyncio.gather(*part), await asyncio.gather(*fin)
    got_p, got_f = asyncio.run(run())
    assert list(got_p) == want_p and list(got_f) == want_f
    assert all(r.batches > 0 for r in two._dispatcher.replicas)
    assert two.transcribe_batch(wavs[:4], max_new_tokens=[8] * 4) == one.transcribe_batch(wavs[:4], max_new_tokens=[8] * 4)
    one.close(); two.close()


@pytest.mark.parametrize("seconds", [65.0, 305.0, 655.0])
def test_long_requests_vs_oracle(orc, seconds):
    """Requests of many 30 s windows behind one prompt (the reference processor accepts up to 655 s = 21 windows, 7883 prompt tokens):
    engine (max_ctx 8192) against the multi-window oracle, TINY dimensions."""
    from sonicscribe_amd import frontend
    from sonicscribe_amd.engine import Engine
    d = replace(spec.TINY, eos_ids=())
    e = Engine(d, 0, max_batch=32, max_ctx=8192)
    e.load_synthetic(5)
    om = orc.Model(d, synth.synth_state_dict(d, 5, bf16=True), bf16=True)
    n = int(seconds * 16000)
    pcm = frontend.normalise_to_int16(synth.synth_pcm(9, n).astype(np.float32) / 32768.0)
    wins = [pcm[s:t] for s, t in frontend.split_windows(len(pcm), d)]
    n_audio, _ = frontend.request_audio_tokens(n, d)
    prompt = [1, 17, 23, 5] + [d.audio_token_id] * n_audio + [7, 301, 302, 9]
    ids, logits = e.transcribe_batch(wins, [prompt], [6], req_win=[0, len(wins)], want_logits=True)
    fm = [orc.logmel(w) for w in wins]
    ref = om.transcribe(np.stack([f for f, _ in fm]), [int(m.sum()) for _, m in fm], prompt, 6)
    assert len(wins) == min(21, -(-n // 480000))
    assert np.abs(logits[:, 0] - ref["step_logits"]).max() <= 4 * 2.0 ** -6
    srt = np.sort(ref["step_logits"], axis=1)
    if (srt[:, -1] - srt[:, -2]).min() > 8 * 2.0 ** -6:
        assert np.array_equal(ids[0], ref["new_ids"])
    e.close()


def test_deep_context_full_depth_vs_oracle(orc):
    """The bench decodes 150 tokens (context 260 -> 410); the full-depth test above stops after 3.  Here the B = 32 x 20 s batch runs all
    150 steps under teacher forcing (varying ids per row) and one row is checked against the bf16 oracle at full depth at every step:
    the later KV slices of the decode attention (contexts past 384 = a fourth 128-key round), token positions / kv_len up to step 149,
    and the embedding gather of 149 different forced ids.  Bound as above: 8 bf16 ulp of 2^-6; measured values are printed."""
    from sonicscribe_amd.engine import Engine
    d = replace(spec.FULL, eos_ids=())
    B, n_new, n_samples, row = 32, 150, 320000, 5
    e = Engine(d, 0, max_batch=B, max_ctx=512)
    e.load_synthetic(SEED)
    segs = [synth.synth_pcm(200 + i, n_samples) for i in range(B)]
    prompt = tiny_prompt(n_samples, d)
    rng = np.random.default_rng(77)
    force = rng.integers(2, d.vocab, size=(B, n_new)).astype(np.int32)
    force[force == d.audio_token_id] = 17
    e.set_forced_ids(force)
    ids, logits = e.transcribe_batch(segs, [prompt] * B, [n_new] * B, want_logits=True)
    e.set_forced_ids(None)
    e.close()
    assert all(np.array_equal(ids[r], force[r]) for r in range(B)) and np.isfinite(logits).all()
    om = orc.Model(d, _full_state(orc, d, SEED), bf16=True)
    feats, mask = orc.logmel(segs[row])
    ref = om.transcribe(feats, int(mask.sum()), prompt, n_new, force_ids=force[row])
    dl = np.abs(logits[:, row] - ref["step_logits"]).max(axis=1)
    tol = 8 * 2.0 ** -6
    print("deep context, full depth: max|dlogit| per step at 0/1/16/17/64/128/149:", [round(float(dl[s]), 4) for s in (0, 1, 16, 17, 64, 128, 149)],
          f"worst {dl.max():.4f} at step {int(dl.argmax())}")
    assert dl.max() <= tol, (float(dl.max()), int(dl.argmax()))
    for s in (1, 16, 17, 64, 149):
        srt = np.sort(ref["step_logits"][s])
        if srt[-1] - srt[-2] > 2 * tol:
            assert int(np.argmax(logits[s, row])) == int(np.argmax(ref["step_logits"][s])), s


def test_stage_entry_points_and_memory_info(eng):
    """sonic_prefill + sonic_decode_step (SURVEY 8b stage entry points) = sonic_run_staged, token for token; the memory / device queries
    behind get_model_info() and the debug dict (asr.py:453-457, 501-506)."""
    from sonicscribe_amd.engine import device_info
    d = spec.TINY
    segs = [synth.synth_pcm(70 + i, 48000 + 16000 * i) for i in range(3)]
    prompts = [tiny_prompt(len(s), d) for s in segs]
    budgets = [9, 20, 5]
    want, _ = eng.transcribe_batch(segs, prompts, budgets)
    eng.stage_pcm(segs)
    eng.prefill(prompts, budgets)
    first = eng.fetch_tokens(3, 32)
    assert [len(x) for x in first] == [1, 1, 1] and all(first[r][0] == want[r][0] for r in range(3))
    total = 0
    while True:
        n_active, done = eng.decode_step(4)
        total += done
        if done == 0:
            break
    got = eng.fetch_tokens(3, 32)
    assert total == max(len(w) for w in want) - 1 or total == max(budgets) - 1
    assert all(np.array_equal(got[r], want[r]) for r in range(3))
    alloc, reserved = eng.memory_info()
    assert reserved >= alloc > eng.weight_bytes() > 0
    di = device_info(0)
    assert di["total_bytes"] > 2 ** 36 and di["free_bytes"] > 0 and di["hip_runtime_version"] > 0 and len(di["name"]) > 0

