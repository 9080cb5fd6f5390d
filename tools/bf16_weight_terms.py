#!/usr/bin/env python3
"""Development probe (CPU only, the oracle): which WEIGHTS, rounded to bf16, move the fp32 logits by how much?
Teacher-forced logits over prompt + 16 tokens of one window, fp32 arithmetic throughout, against the all-fp32 oracle:
  enc    : encoder matrices rounded to bf16, decoder fp32            -- the term a bf16 encoder cannot avoid
  dec    : decoder matrices (and the tied output projection) rounded, encoder fp32
  both   : everything rounded                                         -- the floor of any bf16-weight engine
  enc+dec16: encoder rounded, decoder kept to 16 mantissa bits (what hi + lo bf16 weight planes would hold)
The contract asks for logits within 1e-2 of the fp32 evaluation: if `enc` alone is close to that, no amount of extra
precision in the decoder brings a bf16 engine under it.
Usage: python tools/bf16_weight_terms.py [model ...]   (default: tiny.en base small medium large-v3-turbo)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_prompt  # noqa: E402
from oracle import whisper_oracle as O  # noqa: E402
from tests.helpers import synth_audio  # noqa: E402
from whisper_amd.config import dims_for  # noqa: E402
from whisper_amd.weights import synthetic_state_dict  # noqa: E402


def rounded(sd, part, bits16=False):
    """fp32 state dict with the matrices of `part` ("encoder" / "decoder") rounded to bf16 -- or, bits16, to a hi + lo
    pair of bf16 values (about 16 mantissa bits) -- as an engine would hold them."""
    out = {}
    for k, v in sd.items():
        is_matrix = v.dim() >= 2 and "embed_positions" not in k and "embed_tokens" not in k
        if is_matrix and f"model.{part}." in k:
            hi = v.to(torch.bfloat16).to(torch.float32)
            out[k] = hi + (v - hi).to(torch.bfloat16).to(torch.float32) if bits16 else hi
        else:
            out[k] = v
    if part == "decoder":      # the tied output projection is a decoder product; the embedding LOOK-UP stays fp32
        e = sd["model.decoder.embed_tokens.weight"]
        hi = e.to(torch.bfloat16).to(torch.float32)
        out["proj_out.weight"] = hi + (e - hi).to(torch.bfloat16).to(torch.float32) if bits16 else hi
    return out


def main():
    models = sys.argv[1:] or ["tiny.en", "base", "small", "medium", "large-v3-turbo"]
    print(f"{'model':16s} {'enc':>9s} {'dec':>9s} {'both':>9s} {'enc+dec16':>10s}   (max |dlogit| vs the fp32 oracle, T = prompt + 16)")
    for model in models:
        dims = dims_for(model)
        sd = synthetic_state_dict(dims, seed=0)
        prompt = make_prompt(0) if dims.multilingual else [50257, 50358, 50363, 2425]
        seq = list(prompt) + np.random.default_rng(3).integers(1000, 40000, size=16).tolist()
        mel = O.log_mel_segment(synth_audio(1234), dims.n_mels)
        om = O.OracleModel(sd, dims.n_heads)
        enc = om.encode(mel)
        ref = om.decode_full(seq, enc)[0]
        sd_e = rounded(sd, "encoder")
        enc_r = O.OracleModel(sd_e, dims.n_heads).encode(mel)
        t_enc = np.abs(om.decode_full(seq, enc_r)[0] - ref).max()
        om_d = O.OracleModel(rounded(sd, "decoder"), dims.n_heads)
        t_dec = np.abs(om_d.decode_full(seq, enc)[0] - ref).max()
        t_both = np.abs(om_d.decode_full(seq, enc_r)[0] - ref).max()
        om_16 = O.OracleModel(rounded(sd, "decoder", bits16=True), dims.n_heads)
        t_16 = np.abs(om_16.decode_full(seq, enc_r)[0] - ref).max()
        print(f"{model:16s} {t_enc:9.2e} {t_dec:9.2e} {t_both:9.2e} {t_16:10.2e}", flush=True)


if __name__ == "__main__":
    main()
