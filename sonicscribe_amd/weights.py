"""Checkpoint plumbing: HF safetensors / config.json -> engine tensors (asr.py:120-146 replacement).

PyTorch is used here only to read bf16 safetensors into host memory; nothing on the request
path touches torch.
"""
from __future__ import annotations

import glob
import json
import os
from dataclasses import replace
from typing import Dict, Iterator, Tuple

import numpy as np

from .spec import FULL, ModelDims, tensor_inventory

# on-disk prefix -> in-memory prefix (HF:conversion_mapping.py:610-615)
_RENAMES = (
    ("audio_tower.", "model.audio_tower."),
    ("multi_modal_projector.", "model.multi_modal_projector."),
    ("language_model.model.", "model.language_model."),
    ("language_model.lm_head.", "lm_head."),
)


def canonical_name(name: str) -> str:
    """On-disk tensor name -> module name.  The rules are transformers' own load-time renamings for this model family
    (conversion_mapping.py "qwen2_audio": ^language_model.model -> model.language_model, ^language_model.lm_head -> lm_head,
    ^audio_tower / ^multi_modal_projector -> model.*).  transformers 5.x `save_pretrained` writes the decoder as
    `language_model.model.model.*` (its reverse mapping applied on top of the legacy prefix) and loads that back without missing keys,
    so the doubled `model.` is accepted here as well (tests/test_hf_checkpoint_layout.py)."""
    if not (name.startswith("model.") or name.startswith("lm_head.")):
        for src, dst in _RENAMES:
            if name.startswith(src):
                name = dst + name[len(src):]
                break
    while name.startswith("model.language_model.model."):
        name = "model.language_model." + name[len("model.language_model.model."):]
    return name


def dims_from_config(cfg: Dict) -> ModelDims:
    """Override the defaults with a checkpoint's config.json (GlmAsrConfig layout)."""
    a = cfg.get("audio_config") or {}
    t = cfg.get("text_config") or {}
    d = FULL
    rope_a = a.get("rope_parameters") or {}
    rope_t = t.get("rope_parameters") or {}
    eos = t.get("eos_token_id", cfg.get("eos_token_id", list(d.eos_ids)))
    if isinstance(eos, int):
        eos = [eos]
    hidden = t.get("hidden_size", d.dec_d)
    heads = t.get("num_attention_heads", d.dec_heads)
    return replace(
        d,
        n_mels=a.get("num_mel_bins", d.n_mels),
        enc_d=a.get("hidden_size", d.enc_d), enc_ff=a.get("intermediate_size", d.enc_ff),
        enc_layers=a.get("num_hidden_layers", d.enc_layers), enc_heads=a.get("num_attention_heads", d.enc_heads),
        enc_rope_theta=float(rope_a.get("rope_theta", d.enc_rope_theta)),
        enc_partial_rotary=float(rope_a.get("partial_rotary_factor", a.get("partial_rotary_factor", d.enc_partial_rotary))),
        dec_d=hidden, dec_ff=t.get("intermediate_size", d.dec_ff), dec_layers=t.get("num_hidden_layers", d.dec_layers),
        dec_heads=heads, dec_kv_heads=t.get("num_key_value_heads", d.dec_kv_heads),
        dec_head_dim=t.get("head_dim") or hidden // heads,
        dec_rope_theta=float(rope_t.get("rope_theta", t.get("rope_theta", d.dec_rope_theta))),
        dec_rms_eps=float(t.get("rms_norm_eps", d.dec_rms_eps)),
        vocab=t.get("vocab_size", d.vocab), audio_token_id=cfg.get("audio_token_id", d.audio_token_id),
        eos_ids=tuple(int(x) for x in eos),
    )


def load_dims(checkpoint_dir: str) -> ModelDims:
    p = os.path.join(checkpoint_dir, "config.json")
    if not os.path.exists(p):
        raise FileNotFoundError(f"{p} not found")
    with open(p) as f:
        return dims_from_config(json.load(f))


def iter_safetensors(checkpoint_dir: str) -> Iterator[Tuple[str, np.ndarray, bool]]:
    """Yield (canonical name, array, is_bf16_bits).  bf16 tensors come back as uint16 bit patterns."""
    import torch
    from safetensors import safe_open
    files = sorted(glob.glob(os.path.join(checkpoint_dir, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no .safetensors files under {checkpoint_dir}")
    for fn in files:
        with safe_open(fn, framework="pt", device="cpu") as f:
            for key in f.keys():
                t = f.get_tensor(key)
                name = canonical_name(key)
                if t.dtype == torch.bfloat16:
                    yield name, t.contiguous().view(torch.uint16).numpy(), True
                else:
                    yield name, t.to(torch.float32).contiguous().numpy(), False


def load_checkpoint(engine, checkpoint_dir: str) -> None:
    want = {n: s for n, s, _ in tensor_inventory(engine.dims)}
    seen = set()
    for name, arr, is_bits in iter_safetensors(checkpoint_dir):
        if name == "lm_head.weight":   # tied to embed_tokens (modeling_glmasr.py:517)
            continue
        if name not in want:
            continue
        engine.load_tensor(name, arr, bf16_bits=is_bits)
        seen.add(name)
    missing = [n for n in want if n not in seen]
    if missing:
        raise RuntimeError(f"checkpoint is missing {len(missing)} tensors, e.g. {missing[:3]}")
    engine.finalize()


def save_synthetic_checkpoint(path: str, dims: ModelDims, seed: int) -> None:
    """Write a synthetic checkpoint in the on-disk HF layout (tests of the loader)."""
    import torch
    from safetensors.torch import save_file
    from .synth import synth_state_dict
    os.makedirs(path, exist_ok=True)
    sd = synth_state_dict(dims, seed, bf16=True)
    out = {}
    for k, v in sd.items():
        disk = k
        for src, dst in _RENAMES:
            if k.startswith(dst):
                disk = src + k[len(dst):]
                break
        out[disk] = torch.from_numpy(v.copy()).to(torch.bfloat16)
    save_file(out, os.path.join(path, "model.safetensors"))
    cfg = {
        "model_type": "glmasr", "audio_token_id": dims.audio_token_id,
        "audio_config": {"model_type": "glmasr_encoder", "hidden_size": dims.enc_d, "intermediate_size": dims.enc_ff,
                         "num_hidden_layers": dims.enc_layers, "num_attention_heads": dims.enc_heads, "num_mel_bins": dims.n_mels},
        "text_config": {"model_type": "llama", "vocab_size": dims.vocab, "hidden_size": dims.dec_d, "intermediate_size": dims.dec_ff,
                        "num_hidden_layers": dims.dec_layers, "num_attention_heads": dims.dec_heads,
                        "num_key_value_heads": dims.dec_kv_heads, "head_dim": dims.dec_head_dim, "eos_token_id": list(dims.eos_ids),
                        "rms_norm_eps": dims.dec_rms_eps},
    }
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
