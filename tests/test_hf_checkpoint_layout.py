"""The checkpoint loader against a checkpoint written by the HF library itself: `GlmAsrForConditionalGeneration.save_pretrained` of a
tiny random model (the file layout, tensor names and config.json nesting a real GLM-ASR-Nano-2512 download has), read back by
sonicscribe_amd.weights.  No GLM-ASR checkpoint exists offline; this pins the on-disk contract, not the values of the real weights."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
transformers = pytest.importorskip("transformers")

from sonicscribe_amd import spec, weights      # noqa: E402


def test_loader_reads_what_save_pretrained_writes(tmp_path):
    from transformers import GlmAsrConfig, GlmAsrForConditionalGeneration
    d = spec.TINY
    cfg = GlmAsrConfig(
        audio_config=dict(hidden_size=d.enc_d, intermediate_size=d.enc_ff, num_hidden_layers=d.enc_layers, num_attention_heads=d.enc_heads,
                          num_mel_bins=d.n_mels),
        text_config=dict(vocab_size=d.vocab, hidden_size=d.dec_d, intermediate_size=d.dec_ff, num_hidden_layers=d.dec_layers,
                         num_attention_heads=d.dec_heads, num_key_value_heads=d.dec_kv_heads, head_dim=d.dec_head_dim, eos_token_id=list(d.eos_ids)),
        audio_token_id=d.audio_token_id)
    torch.manual_seed(0)
    model = GlmAsrForConditionalGeneration(cfg).to(torch.bfloat16).eval()
    model.save_pretrained(str(tmp_path), safe_serialization=True)
    assert weights.load_dims(str(tmp_path)) == d                         # config.json as HF nests it -> the engine's dimensions
    want = {n: tuple(s) for n, s, _ in spec.tensor_inventory(d)}
    sd = {k: v for k, v in model.state_dict().items()}
    seen = {}
    for name, arr, is_bits in weights.iter_safetensors(str(tmp_path)):
        seen[name] = (arr, is_bits)
    loaded = {n for n in seen if n in want}
    assert loaded == set(want), (sorted(set(want) - loaded)[:5], sorted(set(seen) - set(want))[:5])
    for n in ("model.audio_tower.conv1.weight", "model.audio_tower.layers.1.self_attn.k_proj.weight", "model.multi_modal_projector.linear_2.bias",
              "model.language_model.layers.0.mlp.gate_proj.weight", "model.language_model.embed_tokens.weight", "model.language_model.norm.weight"):
        arr, is_bits = seen[n]
        assert tuple(arr.shape) == want[n] and is_bits                   # stored bf16, handed over as bit patterns
        assert np.array_equal(arr, sd[n].contiguous().view(torch.uint16).numpy())
    # nothing the engine needs is left to a tensor the file does not have; extras (tied lm_head, rotary buffers) are ignored by name
    extra = set(seen) - set(want)
    assert all(("lm_head" in e) or ("rotary" in e) or ("inv_freq" in e) for e in extra), extra
