"""Model file resolution: the lookup half of urro_whisper/model/download.py (`get_onnx_model_paths`, 217-276).

User-supplied paths win when BOTH files exist (229-237); otherwise the cache directory
`~/.cache/urro_whisper/models/<alias-resolved name, '/' -> '--'>/` is searched for `encoder_model.onnx` and
`decoder_model_merged.onnx` (preferred, 259) or `decoder_model.onnx`.  The reference then downloads what is missing; this
build has no network path on purpose, so missing files raise FileNotFoundError instead."""
from __future__ import annotations

import logging
import os
from typing import Optional, Tuple

logger = logging.getLogger("urro_whisper")

STANDARD_SIZES = ["tiny", "tiny.en", "base", "base.en", "small", "small.en", "medium", "medium.en", "large-v1", "large-v2",
                  "large-v3", "large-v3-turbo"]
SIZE_ALIASES = {"large": "large-v3", "turbo": "large-v3-turbo"}


def cache_model_dir(model_identifier: str, cache_dir: Optional[str] = None) -> str:
    """Directory the reference keeps a model's files in (download.py:240-251)."""
    if cache_dir is None:
        cache_dir = os.path.join(os.path.expanduser("~"), ".cache", "urro_whisper", "models")
    key = model_identifier.lower()
    return os.path.join(cache_dir, SIZE_ALIASES.get(key, key).replace("/", "--"))


def get_onnx_model_paths(model_identifier: str, onnx_encoder_path: Optional[str] = None, onnx_decoder_path: Optional[str] = None,
                         verbose: bool = False, cache_dir: Optional[str] = None) -> Tuple[str, str]:
    if onnx_encoder_path and os.path.isfile(onnx_encoder_path) and onnx_decoder_path and os.path.isfile(onnx_decoder_path):
        logger.info(f"using user-provided onnx paths: encoder='{onnx_encoder_path}', decoder='{onnx_decoder_path}'")
        for path, what in ((onnx_encoder_path, "encoder"), (onnx_decoder_path, "decoder")):
            if not os.path.exists(path + "_data"):
                logger.warning(f"user-provided {what} might be missing data file: {path}_data")
        return onnx_encoder_path, onnx_decoder_path
    model_dir = cache_model_dir(model_identifier, cache_dir)
    logger.info(f"using model cache directory: {os.path.dirname(model_dir)}")
    encoder_path = os.path.join(model_dir, "encoder_model.onnx")
    merged = os.path.join(model_dir, "decoder_model_merged.onnx")
    decoder_path = merged if os.path.exists(merged) else os.path.join(model_dir, "decoder_model.onnx")
    if not os.path.exists(encoder_path) or not os.path.exists(decoder_path):
        raise FileNotFoundError(f"model '{model_identifier}' not found in cache ({model_dir}): expected encoder_model.onnx and "
                                "decoder_model_merged.onnx or decoder_model.onnx there (downloading is not available in this build)")
    logger.info(f"using cached model files: encoder='{encoder_path}', decoder='{decoder_path}'")
    return encoder_path, decoder_path
