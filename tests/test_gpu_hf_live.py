"""The HIP path against the LIVE reference arithmetic, through a checkpoint the HF library wrote.

The reference's arithmetic is transformers' GlmAsrForConditionalGeneration + WhisperFeatureExtractor (backend/asr.py:393-422 calls
them).  The fixtures under tests/golden/ were generated from it in the build container; this test runs it on the GPU
box's CPU at test time instead (transformers is installed there; /root/reference is not needed): a tiny model is written with
`save_pretrained` (the on-disk layout of a real download), loaded by the product's own loader (ASR checkpoint path -> sonic_load_tensor
-> finalize), and greedy-decoded by both sides from the same PCM."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
transformers = pytest.importorskip("transformers")

from sonicscribe_amd import spec, synth, weights      # noqa: E402


def test_live_hf_generate_vs_engine_through_hf_checkpoint(tmp_path):
    from oracle import gen_golden as G                  # the container-side generator: only its model / feature helpers are used
    from sonicscribe_amd.engine import Engine
    d = spec.TINY
    model, _cfg = G.build_tiny(torch.bfloat16)          # GlmAsrForConditionalGeneration with the portable-PRNG weights, bf16
    model.save_pretrained(str(tmp_path), safe_serialization=True)
    assert weights.load_dims(str(tmp_path)) == d
    e = Engine(d, 0, max_batch=2, max_ctx=512)
    weights.load_checkpoint(e, str(tmp_path))           # names as transformers writes them (language_model.model.model.*)
    fe = G.feature_extractor()
    n_new = 12
    tol = 4 * 2.0 ** -6                                  # bf16: 4 ulp at |logit| in [2, 4)
    for seed, n in ((10, 80000), (11, 320000), (12, 20480)):
        pcm = synth.synth_pcm(seed, n)
        feats, mask = G.mel_case(fe, pcm)
        n_audio = spec.audio_token_count(int(mask.sum()))
        ids = G.PROMPT_PREFIX + [d.audio_token_id] * n_audio + G.PROMPT_SUFFIX
        input_ids = torch.tensor([ids], dtype=torch.long)
        with torch.no_grad():
            gen = model.generate(input_ids=input_ids, input_features=torch.from_numpy(feats)[None].to(torch.bfloat16),
                                 input_features_mask=torch.from_numpy(mask)[None].long(), attention_mask=torch.ones_like(input_ids),
                                 max_new_tokens=n_new, do_sample=False, return_dict_in_generate=True, output_logits=True)
        ref_ids = gen.sequences[0, len(ids):].numpy().astype(np.int32)
        ref_logits = torch.stack([l[0] for l in gen.logits]).float().numpy()
        got_ids, got_logits = e.transcribe_batch([pcm], [ids], [n_new], want_logits=True)
        steps = min(len(ref_ids), len(got_ids[0]))
        assert steps >= 1
        for s in range(steps):
            assert np.abs(got_logits[s, 0] - ref_logits[s]).max() <= tol, (seed, s, float(np.abs(got_logits[s, 0] - ref_logits[s]).max()))
            srt = np.sort(ref_logits[s]); margin = srt[-1] - srt[-2]
            if got_ids[0][s] != ref_ids[s]:
                assert margin <= 2 * tol, (seed, s, margin)      # ids are bit-exact outside near-ties
                break                                            # histories diverged on a near-tie: later steps are not comparable
    e.close()
