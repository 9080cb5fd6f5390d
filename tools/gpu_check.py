#!/usr/bin/env python3
"""Development probe: runs each stage of the HIP path against the CPU oracle and prints the differences.
Usage (GPU box): python tools/gpu_check.py [align] [mel] [enc] [dec] [greedy] [--model micro] [--dtype f32]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import whisper_oracle as O  # noqa: E402
from whisper_amd import engine as E  # noqa: E402
from whisper_amd.config import alignment_heads, dims_for  # noqa: E402
from whisper_amd.weights import synthetic_state_dict  # noqa: E402

args = sys.argv[1:]
model = args[args.index("--model") + 1] if "--model" in args else "micro"
dtype = args[args.index("--dtype") + 1] if "--dtype" in args else "f32"
stages = [a for a in args if not a.startswith("--") and a not in (model, dtype)] or ["align", "mel", "enc", "dec", "greedy"]
dev = torch.device("cuda:0")


def synth_audio(seed, n=480000):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    chirp = 0.3 * np.sin(2 * np.pi * (200 * t + (2800 / 60.0) * t * t))
    return (0.1 * rng.standard_normal(n) + chirp).astype(np.float32)


if "align" in stages:
    rng = np.random.default_rng(0)
    B, n_sel, N, S = 3, 4, 70, 1500
    rows = rng.random((B, n_sel, N, S), dtype=np.float32) ** 4
    n_tok = np.array([70, 1, 33], dtype=np.int32)
    cost = E.cost_matrix(torch.from_numpy(rows).to(dev), torch.from_numpy(n_tok)).cpu().numpy()
    for b in range(B):
        att = [rows[b, k, : n_tok[b]][None, None] for k in range(n_sel)]
        ref = O.cost_matrix(att, [(k, 0) for k in range(n_sel)])
        print(f"cost b={b} max abs diff {np.abs(cost[b, : n_tok[b]] - ref).max():.3e}  rel {np.abs(cost[b, : n_tok[b]] / ref - 1).max():.3e}")
    # DTW on the oracle's cost (bit-exact contract)
    N2 = 200
    c = -rng.random((2, N2, S))
    c[1, :, :] = np.round(c[1] * 4) / 4  # many exact ties
    nt = np.array([200, 130], dtype=np.int32)
    bnd, pi, pj, plen = E.dtw(torch.from_numpy(c).to(dev), torch.from_numpy(nt))
    torch.cuda.synchronize()
    for b in range(2):
        i1, i2 = O.dtw_symmetric1(c[b, : nt[b]])
        L = int(plen[b])
        ok = L == len(i1) and np.array_equal(pi[b, :L].cpu().numpy(), i1) and np.array_equal(pj[b, :L].cpu().numpy(), i2)
        bo = O.token_boundaries(i1, i2, int(nt[b]))
        okb = np.array_equal(bnd[b, : nt[b] + 1].cpu().numpy(), bo)
        print(f"dtw b={b} len {L} vs {len(i1)} path equal {ok} boundaries equal {okb}")

sd = dims = eng = None
if any(s in stages for s in ("mel", "enc", "dec", "greedy")):
    dims = dims_for(model)
    sd = synthetic_state_dict(dims, 0)
    key = {"micro": None}.get(model, model)
    heads = alignment_heads(key, dims.dec_layers, dims.n_heads) if key else [(0, 1), (1, 0), (1, 1)]
    t0 = time.time()
    eng = E.Engine(dims, sd, dtype=dtype, max_batch=2, align_heads=heads)
    print(f"engine {model}/{dtype}: {eng.device_bytes / 2**20:.0f} MiB in {time.time() - t0:.1f}s")
    om = O.OracleModel(sd, dims.n_heads)

if "mel" in stages:
    a0, a1 = synth_audio(1), synth_audio(2, 300000)
    audio = np.zeros((2, 480000), np.float32)
    audio[0] = a0
    audio[1, :300000] = a1
    mel = eng.mel(torch.from_numpy(audio).to(dev), torch.tensor([480000, 300000], dtype=torch.int32)).cpu().numpy()
    for b, a in enumerate((a0, a1)):
        ref = O.log_mel_segment(a, dims.n_mels)[0]
        ref64 = O.log_mel_segment(a, dims.n_mels, torch.float64)[0]
        print(f"mel b={b} max|gpu-ref32| {np.abs(mel[b] - ref).max():.3e}  max|gpu-ref64| {np.abs(mel[b] - ref64).max():.3e}  "
              f"max|ref32-ref64| {np.abs(ref - ref64).max():.3e}")

if "enc" in stages or "dec" in stages or "greedy" in stages:
    a0 = synth_audio(1)
    mel_ref = O.log_mel_segment(a0, dims.n_mels)
    enc_ref = om.encode(mel_ref)
    mel_t = torch.from_numpy(np.concatenate([mel_ref, mel_ref * 0.5], 0)).to(dev)
    t0 = time.time()
    enc = eng.encode(mel_t)
    torch.cuda.synchronize()
    print(f"encode 2 windows {time.time() - t0:.3f}s")
    d = np.abs(enc[0].cpu().numpy() - enc_ref[0])
    print(f"enc max abs diff {d.max():.3e} mean {d.mean():.3e} (ref mean |x| {np.abs(enc_ref).mean():.3f})")

if "dec" in stages:
    enc_t = torch.from_numpy(np.concatenate([enc_ref, enc_ref], 0)).to(dev)
    eng.cross_kv(2, enc_t)
    V = dims.vocab
    ids = [3, 17, 900 % V, 5, 321 % V, 77, 12, 1]
    logits_ref, cross_ref = om.decode_full(ids, enc_ref, want_attn=True)
    out = eng.decoder_forward(torch.tensor([ids, ids[::-1]], dtype=torch.int32), logits_all=True, logits_last=True, cross_probs=True)
    torch.cuda.synchronize()
    la = out["logits_all"][0].cpu().numpy()
    print(f"logits_all max abs diff {np.abs(la - logits_ref[0]).max():.3e} (|ref| max {np.abs(logits_ref).max():.3f})")
    print(f"logits_last max abs diff {np.abs(out['logits_last'][0].cpu().numpy() - logits_ref[0, -1]).max():.3e}")
    cp = out["cross_probs"][:, 0].cpu().numpy()
    print(f"cross probs max abs diff {np.abs(cp - np.stack([c[0] for c in cross_ref])).max():.3e}")

if "greedy" in stages:
    enc_t = torch.from_numpy(np.concatenate([enc_ref, enc_ref], 0)).to(dev)
    eng.cross_kv(2, enc_t)
    V = dims.vocab
    sot, eot, ts_begin, ts_end = (V - 200, V - 201, V - 100, V - 1) if model == "micro" else (50258, 50257, 50364, 51864)
    prompts = [[sot, sot + 1, sot + 2, ts_begin], [sot, sot + 1, sot + 2, ts_begin, 11, 12]]
    max_tokens = 28
    t0 = time.time()
    res = eng.decode_greedy(prompts, max_tokens, eot, sot, ts_begin, ts_end)
    torch.cuda.synchronize()
    print(f"greedy {time.time() - t0:.3f}s")
    for b, pr in enumerate(prompts):
        ref_new = O.greedy_reference_schedule(om, enc_ref, pr, max_tokens, eot)
        nn = int(res["n_new"][b])
        got = res["new_tokens"][b, :nn].cpu().tolist()
        print(f"greedy b={b}: tokens equal {got == ref_new} (n={nn} vs {len(ref_new)})")
        if got != ref_new:
            print("  got", got)
            print("  ref", ref_new)
            k = next(i for i, (x, y) in enumerate(zip(got, ref_new)) if x != y)
            lg = om.decode_full(list(pr) + ref_new[:k], enc_ref)[0, -1]
            top = np.argsort(lg)[-3:][::-1]
            print(f"  first mismatch at {k}: oracle top3 {top.tolist()} logits {lg[top].tolist()}")
        text = O.filter_text_tokens(ref_new, sot, ts_begin, ts_end)
        nt = int(res["n_text"][b])
        print(f"  text equal {res['text_ids'][b, :nt].cpu().tolist() == text} n_text {nt} non-text {len(ref_new) - len(text)}")
        if text:
            rows_ref = O.attention_rows_teacher_forced(om, enc_ref, pr, text)
            rows = eng.attn_rows(2, len(text))[b].cpu().numpy()
            ref_sel = np.stack([rows_ref[l][0, h] for l, h in eng.heads])
            print(f"  attention rows max abs diff {np.abs(rows - ref_sel).max():.3e}")
            cost_ref = O.cost_matrix(rows_ref, eng.heads)
            i1, i2 = O.dtw_symmetric1(cost_ref)
            b_ref = O.token_boundaries(i1, i2, len(text))
    bnd = eng.align(res["n_text"])
    torch.cuda.synchronize()
    for b, pr in enumerate(prompts):
        ref_new = O.greedy_reference_schedule(om, enc_ref, pr, max_tokens, eot)
        text = O.filter_text_tokens(ref_new, sot, ts_begin, ts_end)
        if not text:
            continue
        rows_ref = O.attention_rows_teacher_forced(om, enc_ref, pr, text)
        i1, i2 = O.dtw_symmetric1(O.cost_matrix(rows_ref, eng.heads))
        b_ref = O.token_boundaries(i1, i2, len(text))
        got = bnd[b, : len(text) + 1].cpu().numpy()
        print(f"boundaries b={b}: max |frame diff| {np.abs(got - b_ref).max()}  equal {np.array_equal(got, b_ref)}")
print("done")
