"""Weight containers for the transcription path.

The reference loads two ONNX graphs (urro_whisper/model/load.py:13-98) whose initialisers are the
tensors of a HuggingFace `WhisperForConditionalGeneration`; tensors are therefore named as in that
state dict ("model.encoder.layers.0.self_attn.q_proj.weight", ...), fp32, `[out, in]` for linears.

`synthetic_state_dict` is the deterministic generator used by tests and the benchmark (no trained
checkpoint can be fetched): linears and embeddings ~ N(0, 0.02^2), conv kernels ~ N(0, 1/(3*C_in)),
LayerNorm gain 1 + N(0, 0.02^2) and bias N(0, 0.02^2), sinusoidal encoder positions.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Tuple

import numpy as np
import torch

from .config import WhisperDims


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    """Encoder position table: sin | cos halves over log-spaced timescales."""
    if channels % 2 != 0:
        raise ValueError("channels must be even")
    log_inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-log_inc * torch.arange(channels // 2, dtype=torch.float32))
    t = torch.arange(length, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([t.sin(), t.cos()], dim=1)


def _attn_names(prefix: str) -> List[Tuple[str, str]]:
    out = []
    for p in ("q_proj", "k_proj", "v_proj", "out_proj"):
        out.append((f"{prefix}.{p}.weight", "lin"))
        if p != "k_proj":
            out.append((f"{prefix}.{p}.bias", "bias"))
    return out


def tensor_plan(dims: WhisperDims) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, kind) for every tensor of the model, in a fixed order."""
    d, f = dims.d_model, dims.d_ffn
    plan: List[Tuple[str, Tuple[int, ...], str]] = []
    e = "model.encoder"
    plan += [(f"{e}.conv1.weight", (d, dims.n_mels, 3), "conv"), (f"{e}.conv1.bias", (d,), "bias"),
             (f"{e}.conv2.weight", (d, d, 3), "conv"), (f"{e}.conv2.bias", (d,), "bias"),
             (f"{e}.embed_positions.weight", (dims.n_audio_ctx, d), "sinus")]
    for i in range(dims.enc_layers):
        p = f"{e}.layers.{i}"
        for n, k in _attn_names(f"{p}.self_attn"):
            plan.append((n, (d, d) if k == "lin" else (d,), k))
        plan += [(f"{p}.self_attn_layer_norm.weight", (d,), "gain"), (f"{p}.self_attn_layer_norm.bias", (d,), "bias"),
                 (f"{p}.fc1.weight", (f, d), "lin"), (f"{p}.fc1.bias", (f,), "bias"),
                 (f"{p}.fc2.weight", (d, f), "lin"), (f"{p}.fc2.bias", (d,), "bias"),
                 (f"{p}.final_layer_norm.weight", (d,), "gain"), (f"{p}.final_layer_norm.bias", (d,), "bias")]
    plan += [(f"{e}.layer_norm.weight", (d,), "gain"), (f"{e}.layer_norm.bias", (d,), "bias")]
    dd = "model.decoder"
    plan += [(f"{dd}.embed_tokens.weight", (dims.vocab, d), "lin"),
             (f"{dd}.embed_positions.weight", (dims.n_text_ctx, d), "lin")]
    for i in range(dims.dec_layers):
        p = f"{dd}.layers.{i}"
        for n, k in _attn_names(f"{p}.self_attn"):
            plan.append((n, (d, d) if k == "lin" else (d,), k))
        plan += [(f"{p}.self_attn_layer_norm.weight", (d,), "gain"), (f"{p}.self_attn_layer_norm.bias", (d,), "bias")]
        for n, k in _attn_names(f"{p}.encoder_attn"):
            plan.append((n, (d, d) if k == "lin" else (d,), k))
        plan += [(f"{p}.encoder_attn_layer_norm.weight", (d,), "gain"),
                 (f"{p}.encoder_attn_layer_norm.bias", (d,), "bias"),
                 (f"{p}.fc1.weight", (f, d), "lin"), (f"{p}.fc1.bias", (f,), "bias"),
                 (f"{p}.fc2.weight", (d, f), "lin"), (f"{p}.fc2.bias", (d,), "bias"),
                 (f"{p}.final_layer_norm.weight", (d,), "gain"), (f"{p}.final_layer_norm.bias", (d,), "bias")]
    plan += [(f"{dd}.layer_norm.weight", (d,), "gain"), (f"{dd}.layer_norm.bias", (d,), "bias")]
    return plan


def synthetic_state_dict(dims: WhisperDims, seed: int = 0, std: float = 0.02) -> Dict[str, torch.Tensor]:
    """Deterministic fp32 CPU tensors for `dims`; the same (dims, seed) always gives the same bits."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, shape, kind in tensor_plan(dims):
        if kind == "sinus":
            t = sinusoids(shape[0], shape[1])
        elif kind == "conv":
            t = torch.randn(shape, generator=g) * (1.0 / math.sqrt(3 * shape[1]))
        elif kind == "gain":
            t = 1.0 + torch.randn(shape, generator=g) * std
        else:  # "lin" and "bias"
            t = torch.randn(shape, generator=g) * std
        sd[name] = t.contiguous()
    return sd


def check_state_dict(dims: WhisperDims, sd: Dict[str, torch.Tensor]) -> None:
    for name, shape, _ in tensor_plan(dims):
        if name not in sd:
            raise KeyError(f"weight '{name}' missing")
        if tuple(sd[name].shape) != tuple(shape):
            raise ValueError(f"weight '{name}' has shape {tuple(sd[name].shape)}, expected {shape}")


def load_safetensors_dir(path: str) -> Dict[str, torch.Tensor]:
    """Loads every *.safetensors file under `path` (HF layout) into one fp32 CPU dict."""
    import glob
    import os
    from safetensors.torch import load_file

    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no .safetensors files under {path}")
    sd: Dict[str, torch.Tensor] = {}
    for f in files:
        for k, v in load_file(f).items():
            sd[k if k.startswith("model.") else f"model.{k}"] = v.float()
    return sd
