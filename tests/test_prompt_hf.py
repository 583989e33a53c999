"""HFPrompt (the real-checkpoint prompt builder of the façade) against the HF processor itself.

No GLM-ASR checkpoint exists in this image, so the tokenizer is a small synthetic one (word-level vocabulary, the reference's
special-token names, a GLM-style chat template); the processor is the real `GlmAsrProcessor` of the installed transformers with the
real `WhisperFeatureExtractor`.  The reference builds its inputs with `processor.apply_chat_template(messages, tokenize=True, ...)`
(backend/asr.py:375-399), which renders the template and then runs `processor(text=..., audio=...)`: that second half - the audio
placeholder expanded to `num_audio_tokens` copies, then tokenised - is what HFPrompt.build restates without the audio, from the
sample count alone.  Pinned here: the audio-token count (one and several 30 s windows) and the complete id sequence.
"""
import numpy as np
import pytest

transformers = pytest.importorskip("transformers")
tokenizers = pytest.importorskip("tokenizers")

from sonicscribe_amd import frontend, spec           # noqa: E402
from sonicscribe_amd.asr import HFPrompt              # noqa: E402

WORDS = ["<unk>", "<|pad|>", "<|user|>", "<|assistant|>", "<|begin_of_audio|>", "<|end_of_audio|>", "<|endoftext|>", "Please", "transcribe",
         "this", "audio", "into", "text", ".", "Pay", "special", "attention", "to", "these", "important", "terms", ":", '"', ",", "kubernetes", "grafana"]
TEMPLATE = ("{% for m in messages %}<|user|>{% for c in m['content'] %}{% if c['type'] == 'audio' %}<|begin_of_audio|><|pad|><|end_of_audio|>"
            "{% else %}{{ c['text'] }}{% endif %}{% endfor %}{% endfor %}{% if add_generation_prompt %}<|assistant|>{% endif %}")


@pytest.fixture(scope="module")
def processor():
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast, WhisperFeatureExtractor
    from transformers.models.glmasr.processing_glmasr import GlmAsrProcessor
    tk = Tokenizer(models.WordLevel({w: i for i, w in enumerate(WORDS)}, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.WhitespaceSplit(), pre_tokenizers.Punctuation()])
    specials = [w for w in WORDS if w.startswith("<|")]
    tok = PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="<unk>", pad_token="<|endoftext|>", eos_token="<|endoftext|>",
                                  additional_special_tokens=specials)
    return GlmAsrProcessor(WhisperFeatureExtractor(feature_size=128), tok, chat_template=TEMPLATE)


@pytest.mark.parametrize("seconds,hotwords", [(5.0, None), (20.0, None), (1.28, None), (20.0, ["Kubernetes", "grafana"]), (47.5, None), (61.0, ["grafana"])])
def test_hfprompt_matches_processor(processor, seconds, hotwords):
    dims = spec.FULL
    n = int(seconds * 16000)
    pcm = (np.random.default_rng(int(seconds * 10)).standard_normal(n) * 0.1).astype(np.float32)
    instruction = frontend.build_instruction(hotwords)
    messages = [{"role": "user", "content": [{"type": "audio", "url": ""}, {"type": "text", "text": instruction}]}]
    text = processor.tokenizer.apply_chat_template(messages, tokenize=False, add_generation_prompt=True, chat_template=TEMPLATE)
    out = processor(text=text, audio=[pcm], return_tensors="pt")
    want = out["input_ids"][0].tolist()
    n_audio, per_window = frontend.request_audio_tokens(n, dims)
    assert n_audio == int((np.asarray(want) == processor.audio_token_id).sum())           # HF's count, one or several windows
    assert len(per_window) == out["input_features"].shape[0] == len(frontend.split_windows(n, dims))
    prompt = HFPrompt(processor, dims)
    assert prompt.build(instruction, n_audio) == want
    assert prompt.build(instruction, n_audio) is prompt.build(instruction, n_audio)        # cached per (instruction, n_audio)
