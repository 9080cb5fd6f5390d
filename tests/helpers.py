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


def bf16_weights(sd):
    """The weights a bf16 engine computes with, as an fp32 state dict for the oracle: every matrix and conv kernel
    rounded to bf16 (round-to-nearest-even, as the engine packs them); biases, LayerNorm parameters and position tables
    stay fp32, and so does the token embedding where it is LOOKED UP -- only its use as the output projection is bf16,
    hence the separate "proj_out.weight".  An fp32 evaluation on these weights isolates what the bf16 path adds on top
    of the weight format itself."""
    import torch
    out = {}
    for k, v in sd.items():
        is_matrix = v.dim() >= 2 and "embed_positions" not in k and "embed_tokens" not in k
        out[k] = v.to(torch.bfloat16).to(torch.float32) if is_matrix else v
    out["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"].to(torch.bfloat16).to(torch.float32)
    return out


MICRO_SPECIALS = dict(sot=824, eot=823, ts_begin=924, ts_end=1023)       # vocab 1024
WHISPER_SPECIALS = dict(sot=50258, eot=50257, ts_begin=50364, ts_end=51864)


class WordTokenizer:
    """Whisper-tokenizer-shaped stand-in over the multilingual id layout (no tokenizer files exist offline).
    Text ids: 11 ',', 12 ' -', 13 ' >', 14 '.', every other id t < 50257 is the word ' w<t>'.  Used identically by the
    golden generator (which drives the reference's own core.py) and by the tests of this package's host logic."""
    eot, sot, sot_prev, no_timestamps, timestamp_begin = 50257, 50258, 50361, 50363, 50364
    transcribe, translate, task = 50359, 50358, 50359
    language = "en"
    _PUNCT = {11: ",", 12: " -", 13: " >", 14: "."}

    def decode(self, ids):
        out = []
        for t in ids:
            t = int(t)
            if t in self._PUNCT:
                out.append(self._PUNCT[t])
            elif t < self.eot:
                out.append(f" w{t}")
        return "".join(out)

    def encode(self, text, allowed_special=None):
        import re
        specials = {"<|en|>": 50259, "<|transcribe|>": self.transcribe, "<|translate|>": self.translate,
                    "<|startofprev|>": self.sot_prev, "<|notimestamps|>": self.no_timestamps}
        ids = []
        for m in re.finditer(r"<\|[^|]*\|>|w(\d+)|,|-|>|\.", text):
            tok = m.group(0)
            if tok in specials:
                ids.append(specials[tok])
            elif tok.startswith("<|") and tok.endswith("|>"):
                ids.append(self.timestamp_begin + int(round(float(tok[2:-2]) / 0.02)))     # <|t.tt|>
            elif m.group(1):
                ids.append(int(m.group(1)))
            else:
                ids.append({",": 11, "-": 12, ">": 13, ".": 14}[tok])
        return ids
