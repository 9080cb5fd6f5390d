"""Slaney-scale triangular mel filterbank for 16 kHz / n_fft 400, the table whisper.log_mel_spectrogram
loads from its mel_filters.npz (urro_whisper/audio/load.py:83 reaches it through openai-whisper)."""
from __future__ import annotations

import math

import numpy as np

from .config import N_FFT, N_FREQ, SAMPLE_RATE

_F_SP = 200.0 / 3.0
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP
_LOGSTEP = math.log(6.4) / 27.0


def _hz_to_mel(hz: float) -> float:
    return _MIN_LOG_MEL + math.log(hz / _MIN_LOG_HZ) / _LOGSTEP if hz >= _MIN_LOG_HZ else hz / _F_SP


def _mel_to_hz(mel: np.ndarray) -> np.ndarray:
    lin = mel * _F_SP
    log = _MIN_LOG_HZ * np.exp(_LOGSTEP * (mel - _MIN_LOG_MEL))
    return np.where(mel >= _MIN_LOG_MEL, log, lin)


def mel_filterbank(n_mels: int) -> np.ndarray:
    """f32 [n_mels, 201]; each triangle is scaled by 2 / (its band width in Hz)."""
    bins = np.arange(N_FREQ, dtype=np.float64) * (SAMPLE_RATE / N_FFT)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(SAMPLE_RATE / 2.0), n_mels + 2))
    bank = np.zeros((n_mels, N_FREQ), dtype=np.float64)
    for m in range(n_mels):
        lo, mid, hi = edges[m], edges[m + 1], edges[m + 2]
        rise = (bins - lo) / (mid - lo)
        fall = (hi - bins) / (hi - mid)
        bank[m] = np.clip(np.minimum(rise, fall), 0.0, None) * (2.0 / (hi - lo))
    return bank.astype(np.float32)
