"""Shared test inputs.  Everything is regenerated from integer seeds with IEEE-exact operations."""
import numpy as np


def synth_attention(seed, n_layers, n_heads, n_tok, n_frames=1500):
    """Same generator as tools/make_golden.py: per layer f32[1,H,N,S], u^4 of uniform u."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_layers):
        u = rng.random((1, n_heads, n_tok, n_frames), dtype=np.float32)
        u = u * u
        out.append((u * u).astype(np.float32))
    return out


def synth_audio(seed, n=480000):
    """0.1 * N(0,1) noise plus a 200 -> 3000 Hz chirp, 16 kHz."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    chirp = 0.3 * np.sin(2 * np.pi * (200 * t + (2800 / 60.0) * t * t))
    return (0.1 * rng.standard_normal(n) + chirp).astype(np.float32)


MICRO_SPECIALS = dict(sot=824, eot=823, ts_begin=924, ts_end=1023)       # vocab 1024
WHISPER_SPECIALS = dict(sot=50258, eot=50257, ts_begin=50364, ts_end=51864)
