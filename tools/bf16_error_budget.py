#!/usr/bin/env python3
"""Development probe: where does the bf16 path's logit error against the fp32 oracle come from?
Teacher-forced logits over prompt + a few tokens for one window, four ways:
  (a) bf16 engine vs fp32 oracle                       -- what the parity contract measures
  (b) fp32 engine on bf16-rounded weights vs fp32 oracle on the original weights   -- weight rounding alone
  (c) bf16 engine vs fp32 oracle on bf16-rounded weights                            -- activation rounding alone
  (d) as (a) with the encoder states taken from the fp32 oracle                     -- decoder alone
Usage (GPU box): python tools/bf16_error_budget.py [--model small] [--prompt 170]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_prompt  # noqa: E402
from oracle import whisper_oracle as O  # noqa: E402
from tests.helpers import synth_audio  # noqa: E402
from whisper_amd import engine as E  # noqa: E402
from whisper_amd.config import dims_for  # noqa: E402
from whisper_amd.weights import synthetic_state_dict  # noqa: E402

args = sys.argv[1:]
model = args[args.index("--model") + 1] if "--model" in args else "small"
P = int(args[args.index("--prompt") + 1]) if "--prompt" in args else 170
dims = dims_for(model)
sd = synthetic_state_dict(dims, seed=0)
sd_r = {k: v.to(torch.bfloat16).to(torch.float32) if (v.dim() >= 2 and "embed_positions" not in k) else v for k, v in sd.items()}
prompt = make_prompt(0)[:P] if dims.multilingual else [50257, 50358, 50363, 2425]
rng = np.random.default_rng(3)
seq = list(prompt) + rng.integers(1000, 40000, size=16).tolist()
a = synth_audio(1234)
mel = O.log_mel_segment(a, dims.n_mels)
dev = "cuda:0"


def gpu_logits(weights, dtype, enc_override=None):
    eng = E.Engine(dims, weights, dtype=dtype, max_batch=1)
    try:
        enc = eng.encode(torch.from_numpy(mel).to(dev))
        if enc_override is not None:
            eng.cross_kv(1, torch.from_numpy(enc_override).to(dev))
        else:
            eng.cross_kv(1)
        lg = eng.decoder_forward(torch.tensor([seq], dtype=torch.int32), logits_all=True, logits_last=False)["logits_all"][0].cpu().numpy()
        return enc.cpu().numpy(), lg
    finally:
        eng.close()


om, om_r = O.OracleModel(sd, dims.n_heads), O.OracleModel(sd_r, dims.n_heads)
enc_ref, enc_ref_r = om.encode(mel), om_r.encode(mel)
ref, ref_r = om.decode_full(seq, enc_ref)[0], om_r.decode_full(seq, enc_ref_r)[0]
enc_b, lg_b = gpu_logits(sd, "bf16")
enc_f, lg_f = gpu_logits(sd_r, "f32")
_, lg_d = gpu_logits(sd, "bf16", enc_ref)


def rep(name, x, y):
    d = np.abs(x - y)
    print(f"{name:62s} max {d.max():.3e}  rms {np.sqrt((d * d).mean()):.3e}")


print(f"model {model}, T = {len(seq)}, |logit| rms {np.sqrt((ref * ref).mean()):.3f}, |enc| rms {np.sqrt((enc_ref * enc_ref).mean()):.3f}")
rep("(a) bf16 engine vs fp32 oracle: logits", lg_b, ref)
rep("(b) weight rounding alone (f32 engine, rounded weights): logits", lg_f, ref)
rep("    oracle(rounded weights) vs oracle: logits", ref_r, ref)
rep("(c) activation rounding alone (bf16 engine vs rounded oracle)", lg_b, ref_r)
rep("(d) bf16 decoder on fp32-oracle encoder states vs fp32 oracle", lg_d, ref)
rep("    encoder states: bf16 engine vs fp32 oracle", enc_b, enc_ref)
rep("    encoder states: rounded-weight oracle vs fp32 oracle", enc_ref_r, enc_ref)
rep("    encoder states: bf16 engine vs rounded-weight oracle", enc_b, enc_ref_r)

# ---- the KV-cached step path against the same-weights oracle (rms and max per step) ----
from tests.helpers import bf16_weights  # noqa: E402
om_w = O.OracleModel(bf16_weights(sd), dims.n_heads)
enc_w = om_w.encode(mel)
ref_w = om_w.decode_full(seq, enc_w)[0]
eng = E.Engine(dims, sd, dtype="bf16", max_batch=1)
eng.encode(torch.from_numpy(mel).to(dev))
eng.cross_kv(1)
n_pre = len(seq) - 16
lg_multi = eng.decoder_forward(torch.tensor([seq], dtype=torch.int32), logits_all=True, logits_last=False)["logits_all"][0].cpu().numpy()
rep("multi-token pass (split operands) vs same-weights oracle", lg_multi, ref_w)
last = eng.decoder_forward(torch.tensor([seq[:n_pre]], dtype=torch.int32), logits_last=True)["logits_last"][0].cpu().numpy()
rep("prompt-last logits vs same-weights oracle", last, ref_w[n_pre - 1])
steps = []
for t in range(n_pre, len(seq)):
    steps.append(eng.decode_step(torch.tensor([seq[t]]), torch.tensor([t])).cpu().numpy()[0])
steps = np.stack(steps)
rep("16 cached steps vs same-weights oracle", steps, ref_w[n_pre:])
rep("16 cached steps vs multi-token pass (GPU vs GPU)", steps, lg_multi[n_pre:])
eng.close()
