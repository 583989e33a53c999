"""north_star asks for logits "within 1e-3"; a bf16 pipeline cannot be within 1e-3 of ANY other bf16 pipeline (one bf16 ulp at |logit| ~ 3
is 0.0156), the reference's own `mode="native"` path included.  The measurable form of that tolerance: the reference publishes, through
the committed fixture pairs, how far ITS bf16 path is from ITS fp32 path on the same inputs; the HIP path must be no further from that
fp32 truth than the reference's bf16 path is (factor 1.25 on the maximum, and on the mean).  Every committed fixture pair:
TINY free-running prefill, TINY teacher-forced trajectories (2 x 24 steps), the two-window request, and the full-width layers
(vocabulary 59264).  The fp32 fixtures come from transformers in fp32 (oracle/gen_golden.py), not from this repo's arithmetic."""
import os
from dataclasses import replace

import numpy as np
import pytest

from sonicscribe_amd import frontend, spec, synth

pytestmark = pytest.mark.gpu
SEED = 20260128
FACTOR = 1.25


def closer_than_reference_bf16(name, hip, g32, g16):
    ref_max, got_max = float(np.abs(g16 - g32).max()), float(np.abs(hip - g32).max())
    ref_mean, got_mean = float(np.abs(g16 - g32).mean()), float(np.abs(hip - g32).mean())
    print(f"{name}: max|HIP - fp32| {got_max:.4f} vs reference bf16 {ref_max:.4f} (ratio {got_max / ref_max:.2f}); "
          f"mean {got_mean:.5f} vs {ref_mean:.5f} (ratio {got_mean / ref_mean:.2f}); max|HIP - reference bf16| {np.abs(hip - g16).max():.4f}")
    assert got_max <= FACTOR * ref_max, (name, got_max, ref_max)
    assert got_mean <= FACTOR * ref_mean, (name, got_mean, ref_mean)


@pytest.fixture(scope="module")
def eng():
    from sonicscribe_amd.engine import Engine
    e = Engine(spec.TINY, 0, max_batch=4, max_ctx=1024)
    e.load_synthetic(SEED)
    yield e
    e.close()


def test_tiny_prefill_and_forced_steps(eng, golden_dir):
    g32, g16 = np.load(os.path.join(golden_dir, "tiny_fp32.npz")), np.load(os.path.join(golden_dir, "tiny_bf16.npz"))
    f32, f16 = np.load(os.path.join(golden_dir, "tiny_forced_fp32.npz")), np.load(os.path.join(golden_dir, "tiny_forced_bf16.npz"))
    segs, prompts, forces = [], [], []
    for si in range(2):
        p = f"s{si}_"
        assert np.array_equal(f32[p + "force_ids"], f16[p + "force_ids"]) and np.array_equal(f32[p + "prompt_ids"], g32[p + "prompt_ids"])
        segs.append(synth.synth_pcm(int(f32[p + "seg_index"]), int(f32[p + "n_samples"])))
        prompts.append(f32[p + "prompt_ids"]); forces.append(f32[p + "force_ids"])
    force = np.stack(forces)
    eng.set_forced_ids(force)
    try:
        _, logits = eng.transcribe_batch(segs, prompts, [force.shape[1]] * 2, want_logits=True)
    finally:
        eng.set_forced_ids(None)
    for si in range(2):
        p = f"s{si}_"
        closer_than_reference_bf16(f"tiny prefill s{si}", logits[0, si], g32[p + "prefill_logits_last"], g16[p + "prefill_logits_last"])
        closer_than_reference_bf16(f"tiny forced s{si} ({force.shape[1]} steps)", logits[:, si], f32[p + "step_logits"], f16[p + "step_logits"])


def test_tiny_multi_window(eng, golden_dir):
    m32, m16 = np.load(os.path.join(golden_dir, "tiny_multi_fp32.npz")), np.load(os.path.join(golden_dir, "tiny_multi_bf16.npz"))
    assert np.array_equal(m32["force_ids"], m16["force_ids"])
    pcm = synth.synth_pcm(int(m32["seg_index"]), int(m32["n_samples"]))
    wins = [pcm[s:e] for s, e in frontend.split_windows(len(pcm), spec.TINY)]
    force = m32["force_ids"][None]
    eng.set_forced_ids(force)
    try:
        _, logits = eng.transcribe_batch(wins, [m32["prompt_ids"]], [force.shape[1]], req_win=[0, 2], want_logits=True)
    finally:
        eng.set_forced_ids(None)
    closer_than_reference_bf16("tiny two-window request", logits[:, 0], m32["step_logits"], m16["step_logits"])


def test_full_width_vocab_59264(golden_dir):
    from sonicscribe_amd.engine import Engine
    g32, g16 = np.load(os.path.join(golden_dir, "full_fp32.npz")), np.load(os.path.join(golden_dir, "full_bf16.npz"))
    assert np.array_equal(g32["force_ids"], g16["force_ids"])
    d = replace(spec.FULL, enc_layers=1, dec_layers=1)
    e = Engine(d, 0, max_batch=1, max_ctx=512)
    e.load_synthetic(int(g32["seed"]))
    pcm = synth.synth_pcm(int(g32["seg_index"]), int(g32["n_samples"]))
    force = g32["force_ids"][None]
    e.set_forced_ids(force)
    _, logits = e.transcribe_batch([pcm], [g32["prompt_ids"]], [force.shape[1]], want_logits=True)
    e.set_forced_ids(None)
    e.close()
    closer_than_reference_bf16("full-width layers, vocabulary 59264", logits[:, 0, ::16], g32["step_logits_sub"], g16["step_logits_sub"])


def test_full_depth_product_paths_against_the_fp32_kind():
    """Round 6: the fp32 kind of the GPU path (SONIC_MODE_F32, itself within 1e-5 of the reference's fp32 fixtures: tests/test_gpu_fp32_mode.py) makes the fp32
    truth available AT FULL DEPTH - 32 encoder + 28 decoder layers, vocabulary 59264 - where no fixture exists and the CPU oracle needs minutes per segment.
    Same weights (the generator's bf16 values, held as fp32 in the fp32 engine), a 20 s and a 5 s segment in one batch, prefill + 7 teacher-forced steps:
      * the bf16 product path (every production kernel, 60 layers deep) stays within a bf16-sized distance of the fp32 arithmetic: measured max |dlogit| 0.104,
        mean 0.0145 on logits in [-4.8, 4.3]; bounds 0.125 (8 bf16 ulp of 2^-6, the full-depth bound of the oracle tests) and 0.02;
      * its argmax equals the fp32 argmax wherever the fp32 top-1 / top-2 margin exceeds twice that bound;
      * the fp16 kind of the same templates (SONIC_MODE_F16) is 9 x closer, as its rounding is: measured max 0.0115, mean 0.0016; bounds 0.02 / 0.003;
      * the int8 mode (LLM.int8 linears, asr.py:169-210) on the same weights: measured max 0.229, mean 0.036 - the quantisation's distance; bounds 0.5 / 0.06.
    HF semantics: asr.py:407-422 (generate), modeling_glmasr.py:171-346, modeling_llama.py:217-324."""
    from sonicscribe_amd.engine import Engine, MODE_F16, MODE_F32, MODE_INT8, MODE_NATIVE
    d = replace(spec.FULL, eos_ids=())
    segs = [synth.synth_pcm(70, 20 * 16000), synth.synth_pcm(71, 5 * 16000)]
    prompts = [[1, 17, 23, 5] + [d.audio_token_id] * spec.audio_token_count(spec.valid_frames(len(s))) + [7, 301, 302, 303, 9, 11] for s in segs]
    rng = np.random.default_rng(17)
    force = rng.integers(2, 59000, size=(2, 8)).astype(np.int32)
    out = {}
    for tag, mode in (("fp32", MODE_F32), ("bf16", MODE_NATIVE), ("fp16", MODE_F16), ("int8", MODE_INT8)):
        e = Engine(d, 0, mode, max_batch=2, max_ctx=512)
        if mode == MODE_F32:
            e.set_option("f32_synth_bf16", 1)
        e.load_synthetic(SEED)
        e.set_forced_ids(force)
        ids, lg = e.transcribe_batch(segs, prompts, [8, 8], want_logits=True)
        e.set_forced_ids(None)
        assert all(np.array_equal(ids[r], force[r]) for r in range(2))
        out[tag] = lg
        e.close()
    truth = out["fp32"]
    assert np.isfinite(truth).all() and truth.std() > 0.3
    srt = np.sort(truth, axis=-1)
    margin = srt[..., -1] - srt[..., -2]
    # int8 = LLM.int8 linears (asr.py:169-210) on the same weights: its distance is the quantisation's, reported and bounded loosely (parity of that mode is unpinned:
    # bitsandbytes is absent; what is pinned is the restated algorithm, tests/test_gpu_int8.py)
    for tag, bound_max, bound_mean in (("bf16", 0.125, 0.02), ("fp16", 0.02, 0.003), ("int8", 0.5, 0.06)):
        dl = np.abs(out[tag] - truth)
        agree = out[tag].argmax(-1) == truth.argmax(-1)
        print(f"full depth (32 + 28 layers, vocabulary 59264), 2 rows x 8 steps, {tag} product path vs the fp32 kind: max |dlogit| {dl.max():.4f}, mean {dl.mean():.5f}, "
              f"logits in [{truth.min():.2f}, {truth.max():.2f}], argmax equal at {int(agree.sum())} of {agree.size} (row, step) pairs, smallest fp32 margin {margin.min():.4f}")
        assert dl.max() <= bound_max and dl.mean() <= bound_mean, (tag, float(dl.max()), float(dl.mean()))
        assert agree[margin > 2 * bound_max].all()
