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
    from tests import hf_helpers as G                   # model / feature helpers over transformers (shared with oracle/gen_golden.py)
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


def _synthetic_processor(d):
    """A GlmAsrProcessor around a word-level tokenizer whose special ids are the model's (audio placeholder, the three EOS ids)."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast, WhisperFeatureExtractor
    from transformers.models.glmasr.processing_glmasr import GlmAsrProcessor
    specials = {d.audio_token_id: "<|pad|>", d.eos_ids[0]: "<|endoftext|>", d.eos_ids[1]: "<|user|>", d.eos_ids[2]: "<|assistant|>",
                4: "<|begin_of_audio|>", 5: "<|end_of_audio|>", 0: "<unk>"}
    words = iter(["Please", "transcribe", "this", "audio", "into", "text", ".", ":", '"', ",", "Pay", "special", "attention", "to", "these",
                  "important", "terms", "alpha", "beta"])
    vocab = {}
    for i in range(d.vocab):
        vocab[specials[i] if i in specials else (next(words, None) or f"w{i}")] = i
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.WhitespaceSplit(), pre_tokenizers.Punctuation()])
    tok = PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="<unk>", pad_token="<|endoftext|>", eos_token="<|endoftext|>",
                                  additional_special_tokens=[s for s in specials.values() if s != "<unk>"])
    template = ("{% for m in messages %}<|user|>{% for c in m['content'] %}{% if c['type'] == 'audio' %}<|begin_of_audio|><|pad|><|end_of_audio|>"
                "{% else %}{{ c['text'] }}{% endif %}{% endfor %}{% endfor %}{% if add_generation_prompt %}<|assistant|>{% endif %}")
    return GlmAsrProcessor(WhisperFeatureExtractor(feature_size=128), tok, chat_template=template), template


def test_facade_on_a_complete_checkpoint_directory_vs_live_reference_sequence(tmp_path):
    """ASRModel(checkpoint_dir) as models_manager.py:26-32 constructs it, on a directory that holds what a download holds (config.json,
    safetensors, tokenizer, chat template, processor / feature-extractor config - all written by the HF library), against the
    reference's own sequence run live: peak-normalise + PCM_16 round trip (asr.py:247-276), chat-template prompt with the audio
    placeholder expanded by the processor (:375-399), generate(do_sample=False) (:411-422), batch_decode(skip_special_tokens)[0].strip()
    (:425-429).  Transcripts must be equal strings, with and without hotwords, for one- and two-window audio."""
    from tests import hf_helpers as G
    from sonicscribe_amd import frontend
    from sonicscribe_amd.asr import ASRModel, HFPrompt
    d = spec.TINY
    model, _ = G.build_tiny(torch.bfloat16)
    model.save_pretrained(str(tmp_path), safe_serialization=True)
    proc, template = _synthetic_processor(d)
    proc.save_pretrained(str(tmp_path))
    m = ASRModel(str(tmp_path), device="cuda", mode="native", max_batch=4, max_ctx=1024)       # no _allow_synthetic_prompt: the real path
    assert isinstance(m.prompt, HFPrompt) and m.dims == d and m.target_sr == 16000
    for seed, seconds, hot, max_new in ((3, 5.0, None, 12), (4, 20.0, ["Alpha", "beta"], 16), (5, 36.0, None, 10)):
        n = int(seconds * 16000)
        wav = synth.synth_pcm(seed, n).astype(np.float32) / np.float32(32768.0) * np.float32(0.5)      # not peak-normalised on purpose
        got = m.transcribe(torch.from_numpy(wav)[None], sampling_rate=16000, max_new_tokens=max_new, hotwords=hot)
        # the reference sequence, live
        audio = frontend.normalise_to_int16(wav).astype(np.float32) / np.float32(32768.0)      # what HF reads back from the temp WAV
        messages = [{"role": "user", "content": [{"type": "audio", "url": ""}, {"type": "text", "text": frontend.build_instruction(hot)}]}]
        text = proc.tokenizer.apply_chat_template(messages, tokenize=False, add_generation_prompt=True, chat_template=template)
        inputs = proc(text=text, audio=[audio], return_tensors="pt")
        with torch.no_grad():
            gen = model.generate(input_ids=inputs["input_ids"], attention_mask=inputs["attention_mask"],
                                 input_features=inputs["input_features"].to(torch.bfloat16), input_features_mask=inputs["input_features_mask"],
                                 max_new_tokens=max_new, do_sample=False, return_dict_in_generate=True, output_logits=True)
        new = gen.sequences[:, inputs["input_ids"].shape[1]:]
        want = proc.batch_decode(new, skip_special_tokens=True)[0].strip()
        logits = torch.stack([l[0] for l in gen.logits]).float().numpy()
        srt = np.sort(logits, axis=1)
        print(f"{seconds:.0f} s, hotwords {hot}: min margin {(srt[:, -1] - srt[:, -2]).min():.3f}; engine {got!r}; reference {want!r}")
        if (srt[:, -1] - srt[:, -2]).min() > 8 * 2.0 ** -6:       # no near-tie on the reference's trajectory: strings must be identical
            assert got == want, (seconds, got, want)
        else:                                                      # a near-tie may flip one id: the common prefix must still be long
            a, b = got.split(), want.split()
            same = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
            assert same >= 1, (seconds, got, want)
    m.close()
