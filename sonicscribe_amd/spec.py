"""Model dimensions and the canonical tensor inventory of GLM-ASR-Nano.

The numbers follow the defaults of the third-party implementation the reference
calls into: HF:models/glmasr/configuration_glmasr.py:44-54,86-103
for the encoder / decoder sizes and HF:models/glmasr/modeling_glmasr.py:287-346
for the module layout.  Tensor names are the *in-memory* names of
``GlmAsrForConditionalGeneration.state_dict()``; the on-disk names of the
checkpoint (``audio_tower.*`` / ``language_model.model.*``,
HF:conversion_mapping.py:610-615) are mapped onto them by ``weights.py``.

This file is host logic only (no torch, no GPU).
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import Dict, Iterator, List, Tuple


@dataclass(frozen=True)
class ModelDims:
    # --- log-mel front-end (HF:feature_extraction_whisper.py:69-103) ---
    sampling_rate: int = 16000
    n_fft: int = 400
    hop: int = 160
    n_mels: int = 128
    chunk_seconds: int = 30
    # --- encoder (GlmAsrEncoderConfig) ---
    enc_d: int = 1280
    enc_ff: int = 5120
    enc_layers: int = 32
    enc_heads: int = 20
    enc_rope_theta: float = 10000.0
    enc_partial_rotary: float = 0.5
    enc_ln_eps: float = 1e-5
    # --- projector (GlmAsrMultiModalProjector) ---
    merge: int = 4
    # --- decoder (Llama text_config) ---
    dec_d: int = 2048
    dec_ff: int = 6144
    dec_layers: int = 28
    dec_heads: int = 16
    dec_kv_heads: int = 4
    dec_head_dim: int = 128
    dec_rope_theta: float = 10000.0
    dec_rms_eps: float = 1e-5
    vocab: int = 59264
    audio_token_id: int = 59260
    eos_ids: Tuple[int, ...] = (59246, 59253, 59255)

    # ---- derived ----
    @property
    def n_samples(self) -> int:
        return self.chunk_seconds * self.sampling_rate  # 480000

    @property
    def n_frames(self) -> int:
        return self.n_samples // self.hop  # 3000

    @property
    def enc_T(self) -> int:
        return self.n_frames // 2  # 1500 (conv2 stride 2)

    @property
    def enc_head_dim(self) -> int:
        return self.enc_d // self.enc_heads

    @property
    def enc_rotary_dim(self) -> int:
        return int(self.enc_head_dim * self.enc_partial_rotary)

    @property
    def proj_in(self) -> int:
        return self.enc_d * self.merge  # == enc_ff for the shipped config

    @property
    def proj_mid(self) -> int:
        return self.dec_d * 2

    @property
    def max_audio_tokens(self) -> int:
        return self.enc_T // self.merge  # 375

    def to_dict(self) -> Dict:
        d = asdict(self)
        d["eos_ids"] = list(self.eos_ids)
        return d


FULL = ModelDims()

# Tiny configuration used by the golden fixtures (tests/golden) and the smoke test.  Head
# dims (64 encoder / 128 decoder) and the rotary split match the full model so the same
# kernel instantiations are exercised; all GEMM K dims stay multiples of 64.
TINY = ModelDims(
    enc_d=128, enc_ff=512, enc_layers=2, enc_heads=2,
    dec_d=256, dec_ff=512, dec_layers=2, dec_heads=2, dec_kv_heads=1, dec_head_dim=128,
    vocab=1024, audio_token_id=1000, eos_ids=(990, 991, 992),
)


def audio_token_count(n_valid_frames: int, merge: int = 4) -> int:
    """Number of audio placeholder tokens for ``n_valid_frames`` valid mel frames.

    Follows HF:models/glmasr/processing_glmasr.py:128-134 (the two conv length formulas
    followed by the 4-frame merge).
    """
    L = n_valid_frames
    for padding, kernel, stride in ((1, 3, 1), (1, 3, 2)):
        L = (L + 2 * padding - (kernel - 1) - 1) // stride + 1
    return (L - merge) // merge + 1


def valid_frames(n_samples: int, dims: ModelDims = FULL) -> int:
    """Valid mel frames of one <=30 s window: ``attention_mask[::hop]`` summed
    (HF:feature_extraction_whisper.py:332-341)."""
    n = min(n_samples, dims.n_samples)
    return (n + dims.hop - 1) // dims.hop if n > 0 else 0


# kind -> (scale rule, offset) used by the synthetic generator (synth.py)
#   "mat":   U(-s, s), s = sqrt(3 / fan_in)      (unit-gain linear / conv weights)
#   "embed": U(-s, s), s = sqrt(3 / d)           (tied embedding / lm_head)
#   "bias":  U(-0.1, 0.1)
#   "norm":  1 + U(-0.1, 0.1)


def tensor_inventory(d: ModelDims) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, kind) of every parameter, in a fixed canonical order."""
    inv: List[Tuple[str, Tuple[int, ...], str]] = []
    at = "model.audio_tower."
    inv.append((at + "conv1.weight", (d.enc_d, d.n_mels, 3), "mat"))
    inv.append((at + "conv1.bias", (d.enc_d,), "bias"))
    inv.append((at + "conv2.weight", (d.enc_d, d.enc_d, 3), "mat"))
    inv.append((at + "conv2.bias", (d.enc_d,), "bias"))
    for i in range(d.enc_layers):
        p = f"{at}layers.{i}."
        inv.append((p + "input_layernorm.weight", (d.enc_d,), "norm"))
        inv.append((p + "input_layernorm.bias", (d.enc_d,), "bias"))
        inv.append((p + "self_attn.q_proj.weight", (d.enc_d, d.enc_d), "mat"))
        inv.append((p + "self_attn.q_proj.bias", (d.enc_d,), "bias"))
        inv.append((p + "self_attn.k_proj.weight", (d.enc_d, d.enc_d), "mat"))  # no bias (modeling_glmasr.py:184)
        inv.append((p + "self_attn.v_proj.weight", (d.enc_d, d.enc_d), "mat"))
        inv.append((p + "self_attn.v_proj.bias", (d.enc_d,), "bias"))
        inv.append((p + "self_attn.o_proj.weight", (d.enc_d, d.enc_d), "mat"))
        inv.append((p + "self_attn.o_proj.bias", (d.enc_d,), "bias"))
        inv.append((p + "post_attention_layernorm.weight", (d.enc_d,), "norm"))
        inv.append((p + "post_attention_layernorm.bias", (d.enc_d,), "bias"))
        inv.append((p + "mlp.fc1.weight", (d.enc_ff, d.enc_d), "mat"))
        inv.append((p + "mlp.fc1.bias", (d.enc_ff,), "bias"))
        inv.append((p + "mlp.fc2.weight", (d.enc_d, d.enc_ff), "mat"))
        inv.append((p + "mlp.fc2.bias", (d.enc_d,), "bias"))
    inv.append((at + "norm.weight", (d.enc_d,), "norm"))
    inv.append((at + "norm.bias", (d.enc_d,), "bias"))
    pj = "model.multi_modal_projector."
    inv.append((pj + "linear_1.weight", (d.proj_mid, d.proj_in), "mat"))
    inv.append((pj + "linear_1.bias", (d.proj_mid,), "bias"))
    inv.append((pj + "linear_2.weight", (d.dec_d, d.proj_mid), "mat"))
    inv.append((pj + "linear_2.bias", (d.dec_d,), "bias"))
    lm = "model.language_model."
    inv.append((lm + "embed_tokens.weight", (d.vocab, d.dec_d), "embed"))
    qd = d.dec_heads * d.dec_head_dim
    kd = d.dec_kv_heads * d.dec_head_dim
    for i in range(d.dec_layers):
        p = f"{lm}layers.{i}."
        inv.append((p + "input_layernorm.weight", (d.dec_d,), "norm"))
        inv.append((p + "self_attn.q_proj.weight", (qd, d.dec_d), "mat"))
        inv.append((p + "self_attn.k_proj.weight", (kd, d.dec_d), "mat"))
        inv.append((p + "self_attn.v_proj.weight", (kd, d.dec_d), "mat"))
        inv.append((p + "self_attn.o_proj.weight", (d.dec_d, qd), "mat"))
        inv.append((p + "post_attention_layernorm.weight", (d.dec_d,), "norm"))
        inv.append((p + "mlp.gate_proj.weight", (d.dec_ff, d.dec_d), "mat"))
        inv.append((p + "mlp.up_proj.weight", (d.dec_ff, d.dec_d), "mat"))
        inv.append((p + "mlp.down_proj.weight", (d.dec_d, d.dec_ff), "mat"))
    inv.append((lm + "norm.weight", (d.dec_d,), "norm"))
    # lm_head.weight is tied to embed_tokens (modeling_glmasr.py:517) and is not listed.
    return inv


def param_count(d: ModelDims) -> int:
    n = 0
    for _, shape, _ in tensor_inventory(d):
        k = 1
        for s in shape:
            k *= s
        n += k
    return n
