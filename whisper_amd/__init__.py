"""MI355X-native implementation of urro_whisper's transcription hot path (drop-in for `whisperer`)."""
from .core import whisperer  # noqa: F401

__all__ = ["whisperer"]
