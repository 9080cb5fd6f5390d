"""Model shapes, audio constants and alignment-head tables for the transcription path.

Reference constants mirrored here:
  * chunk / sample-rate / hop constants: urro_whisper/core.py:30-34, urro_whisper/align.py:37-46
  * n_mels rule (128 only when "large-v3" is in the model name): urro_whisper/core.py:180
  * alignment heads: flat indices of the boolean masks in urro_whisper/align.py:49-67
    (base85 + gzip dumps there; stored here already decoded as flat `layer * n_heads + head` indices)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
N_FREQ = N_FFT // 2 + 1            # 201 one-sided bins
CHUNK_LENGTH_SAMPLES = 30 * SAMPLE_RATE
N_FRAMES = 3000                    # mel frames fed to the encoder
N_AUDIO_CTX = 1500                 # encoder positions after the stride-2 conv
N_TEXT_CTX = 448                   # decoder positions
HEAD_DIM = 64
AUDIO_TIME_PER_TOKEN = 0.02        # seconds per encoder frame
MEDFILT_WIDTH = 7


@dataclass(frozen=True)
class WhisperDims:
    name: str
    d_model: int
    enc_layers: int
    dec_layers: int
    n_heads: int
    vocab: int
    n_mels: int = 80
    n_audio_ctx: int = N_AUDIO_CTX
    n_text_ctx: int = N_TEXT_CTX

    @property
    def d_ffn(self) -> int:
        return 4 * self.d_model

    @property
    def multilingual(self) -> bool:
        return not self.name.endswith(".en")


_PRESETS: Dict[str, WhisperDims] = {
    "tiny.en": WhisperDims("tiny.en", 384, 4, 4, 6, 51864),
    "tiny": WhisperDims("tiny", 384, 4, 4, 6, 51865),
    "base.en": WhisperDims("base.en", 512, 6, 6, 8, 51864),
    "base": WhisperDims("base", 512, 6, 6, 8, 51865),
    "small.en": WhisperDims("small.en", 768, 12, 12, 12, 51864),
    "small": WhisperDims("small", 768, 12, 12, 12, 51865),
    "medium.en": WhisperDims("medium.en", 1024, 24, 24, 16, 51864),
    "medium": WhisperDims("medium", 1024, 24, 24, 16, 51865),
    "large-v1": WhisperDims("large-v1", 1280, 32, 32, 20, 51865),
    "large-v2": WhisperDims("large-v2", 1280, 32, 32, 20, 51865),
    "large-v3": WhisperDims("large-v3", 1280, 32, 32, 20, 51866, n_mels=128),
    "large-v3-turbo": WhisperDims("large-v3-turbo", 1280, 32, 4, 20, 51866, n_mels=128),
    # a shape small enough for CPU parity tests; not a reference model
    "micro": WhisperDims("micro", 128, 2, 2, 2, 1024),
}
_ALIASES = {"large": "large-v3", "turbo": "large-v3-turbo"}


def dims_for(model: str) -> WhisperDims:
    key = _ALIASES.get(model, model)
    if key.endswith("_timestamped"):
        key = key[: -len("_timestamped")]
    if key.startswith("whisper-"):
        key = key[len("whisper-"):]
    if key not in _PRESETS:
        raise KeyError(f"unknown whisper model shape '{model}'")
    return _PRESETS[key]


def n_mels_for(model: str) -> int:
    """The reference picks the mel count from the model *string*, so the aliases
    "large"/"turbo" get 80 (urro_whisper/core.py:180)."""
    return 128 if "large-v3" in model else 80


# flat indices (layer * n_heads + head) of the heads used for word alignment, with the mask size
ALIGNMENT_HEADS: Dict[str, Tuple[int, List[int]]] = {
    "tiny.en": (24, [6, 12, 17, 18, 19, 20, 21, 22]),
    "tiny": (24, [14, 18, 20, 21, 22, 23]),
    "base.en": (48, [27, 39, 41, 45, 47]),
    "base": (48, [25, 34, 35, 39, 41, 42, 44, 46]),
    "small.en": (144, [78, 84, 87, 92, 98, 101, 103, 108, 112, 116, 118, 120, 121, 122, 123, 126, 131, 134, 136]),
    "small": (144, [63, 69, 96, 100, 103, 104, 108, 115, 117, 125]),
    "medium.en": (384, [180, 225, 236, 238, 244, 256, 260, 265, 284, 286, 295, 298, 303, 320, 323, 329, 334, 348]),
    "medium": (384, [223, 244, 255, 257, 320, 372]),
    "large-v1": (640, [199, 222, 224, 237, 447, 451, 457, 462, 475]),
    "large-v2": (640, [212, 277, 331, 332, 333, 355, 356, 364, 371, 379, 391, 422, 423, 443, 449, 452, 465,
                       467, 473, 505, 521, 532, 555]),
    "large-v3": (640, [140, 217, 258, 272, 321, 354, 391, 424, 481, 506]),
    "large": (640, [140, 217, 258, 272, 321, 354, 391, 424, 481, 506]),
    "large-v3-turbo": (80, [44, 51, 63, 66, 71, 74]),
    "turbo": (80, [44, 51, 63, 66, 71, 74]),
}


def alignment_heads_key(model: str, hidden_dim: int, language: str) -> str:
    """Which mask the reference would pick (urro_whisper/align.py:572-576)."""
    size_map = {384: "tiny", 512: "base", 768: "small", 1024: "medium", 1280: "large"}
    base = size_map.get(hidden_dim, model.split("-")[0].split(".")[0])
    key = f"{base}.en" if (language == "en" and f"{base}.en" in ALIGNMENT_HEADS) else base
    if model in ALIGNMENT_HEADS:
        key = model
    return key


def alignment_heads(key: str, num_layers: int, num_heads: int) -> Optional[List[Tuple[int, int]]]:
    """(layer, head) pairs in row-major mask order, or None when the key is unknown or the mask
    size does not match num_layers * num_heads (urro_whisper/align.py:70-91: no alignment then)."""
    if key not in ALIGNMENT_HEADS:
        return None
    size, flat = ALIGNMENT_HEADS[key]
    if size != num_layers * num_heads:
        return None
    return [(i // num_heads, i % num_heads) for i in flat]
