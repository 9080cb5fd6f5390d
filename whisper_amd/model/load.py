"""Model loading seam (mirror of urro_whisper/model/load.py).

`load_onnx_models(encoder_path, decoder_path, onnx_providers, verbose, exclude_providers)` keeps the reference's
signature and provider-selection rules (load.py:13-53) with one provider: "MI355XExecutionProvider".  Weights
are resolved by `resolve_weights`: a directory of HF-layout *.safetensors files, or "synthetic:<shape>" for the
deterministic generator, or the reference's own file pair (`encoder_model.onnx` + `decoder_model[_merged].onnx`,
urro_whisper/model/download.py:217-276) whose initialisers are read by `onnx_weights` (parity unpinned: no exported
Whisper graph exists offline; that reader is tested on files written by the test-suite)."""
from __future__ import annotations

import logging
import os
from typing import Dict, List, Optional, Tuple

import torch

from ..config import alignment_heads, alignment_heads_key, dims_for
from ..session import PROVIDER, DecoderSession, EncoderSession
from ..weights import load_safetensors_dir, synthetic_state_dict

logger = logging.getLogger("urro_whisper")

AVAILABLE_PROVIDERS = [PROVIDER]


def resolve_weights(model: str, encoder_path: Optional[str] = None, decoder_path: Optional[str] = None
                    ) -> Tuple[str, Dict[str, torch.Tensor]]:
    """-> (shape name, fp32 state dict).  Sources, in this order: "synthetic:<shape>" (deterministic generator); a
    directory of HF-layout *.safetensors files; the reference's own file pair -- the user's `onnx_encoder` /
    `onnx_decoder` when both exist, else the files the reference keeps under ~/.cache/urro_whisper/models/<name>/
    (download.py:217-259; nothing is downloaded here)."""
    src = encoder_path or model
    if isinstance(src, str) and src.startswith("synthetic:"):
        name = src.split(":", 1)[1]
        return name, synthetic_state_dict(dims_for(name), seed=0)
    if isinstance(src, str) and os.path.isdir(src):
        name = model if not os.path.isdir(model) else os.path.basename(os.path.normpath(model))
        return name, load_safetensors_dir(src)
    from ..weights import check_state_dict
    from .download import SIZE_ALIASES, get_onnx_model_paths
    from .onnx_weights import load_onnx_pair
    enc_file, dec_file = get_onnx_model_paths(model, encoder_path, decoder_path)      # raises FileNotFoundError
    sd = load_onnx_pair(enc_file, dec_file)
    if model and not model.endswith(".onnx"):
        name = SIZE_ALIASES.get(model.lower(), model)
    else:
        name = os.path.basename(os.path.dirname(os.path.abspath(enc_file)))
    check_state_dict(dims_for(name), sd)
    return name, sd


def decoder_file_for(model: str, encoder_path: Optional[str] = None, decoder_path: Optional[str] = None) -> Optional[str]:
    """The decoder file resolve_weights would read (None for synthetic / safetensors sources): its name says whether
    the session should look like the merged export."""
    src = encoder_path or model
    if isinstance(src, str) and (src.startswith("synthetic:") or os.path.isdir(src)):
        return None
    from .download import get_onnx_model_paths
    return get_onnx_model_paths(model, encoder_path, decoder_path)[1]


def select_providers(onnx_providers: Optional[List[str]], exclude_providers: Optional[List[str]]) -> List[str]:
    """Provider choice with the reference's exclusion / fallback behaviour (load.py:18-53)."""
    use = list(onnx_providers) if onnx_providers is not None else list(AVAILABLE_PROVIDERS)
    if exclude_providers:
        before = list(use)
        use = [p for p in use if p not in exclude_providers]
        if not use and before:
            logger.warning(f"falling back to providers before exclusion: {before}")
            use = before
        elif not use:
            raise RuntimeError("no suitable onnx execution providers found after exclusion")
    if not use:
        raise RuntimeError("no available onnx execution providers")
    if PROVIDER not in use:
        raise RuntimeError(f"only {PROVIDER} is available in this build (requested {use})")
    return use


def load_onnx_models(encoder_path, decoder_path, onnx_providers=None, verbose=False, exclude_providers=None, *,
                     model: Optional[str] = None, dtype: str = "bf16", max_batch: int = 1, language: str = "en"):
    """-> (encoder_sess, decoder_sess) sharing one engine (one GPU context)."""
    from ..engine import Engine
    select_providers(onnx_providers, exclude_providers)
    name, sd = resolve_weights(model or encoder_path, encoder_path, decoder_path)
    dims = dims_for(name)
    heads = alignment_heads(alignment_heads_key(name, dims.d_model, language), dims.dec_layers, dims.n_heads) or []
    eng = Engine(dims, sd, dtype=dtype, max_batch=max_batch, align_heads=heads)
    # decoder_model_merged.onnx (what the reference prefers, download.py:259) declares the cache inputs; mirror that
    dec_file = decoder_file_for(model or encoder_path, encoder_path, decoder_path) or decoder_path
    merged = isinstance(dec_file, str) and "merged" in os.path.basename(dec_file)
    enc, dec = EncoderSession(eng), DecoderSession(eng, merged=merged)
    logger.info(f"models loaded using {enc.get_providers()} for encoder and {dec.get_providers()} for decoder")
    return enc, dec


def get_tokenizer_for_model(model, language, task):
    """Same contract as the reference (load.py:100-110): openai-whisper's tokenizer, called WITHOUT num_languages."""
    is_multilingual = not model.endswith(".en")
    try:
        from whisper.tokenizer import get_tokenizer
        return get_tokenizer(multilingual=is_multilingual, task=task, language=language), is_multilingual
    except Exception as e:
        raise RuntimeError(f"failed to load tokenizer for model '{model}' (multilingual={is_multilingual}, lang={language}, "
                           f"task={task}), error: {e}")
