#!/usr/bin/env python3
"""Generates tests/golden/*.  Runs only in the build container (needs /root/reference and
HuggingFace transformers); the GPU box and the test-suite read the committed outputs.

Sources of truth used here:
  ref_align_*   urro_whisper/align.py itself, imported by file path with stub `dtw` / `icu` modules
                (both are absent here).  The stub `dtw.dtw` captures the f64 cost matrix the reference
                hands to DTW and returns a path computed by oracle.dtw_symmetric1, so the reference's own
                boundary code (align.py:835-853) runs on it.
  hf_*          transformers' WhisperForConditionalGeneration / WhisperFeatureExtractor with the
                synthetic "micro" weights: the architecture the reference's ONNX files are exported from.

Inputs are regenerated in the tests from integer seeds with IEEE-exact operations only, so that only
the expected outputs are stored.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
GOLD = os.path.join(ROOT, "tests", "golden")

from oracle import whisper_oracle as O  # noqa: E402
from whisper_amd.config import dims_for  # noqa: E402
from whisper_amd.weights import synthetic_state_dict  # noqa: E402


def synth_attention(seed, n_layers, n_heads, n_tok, n_frames=1500):
    """Per layer f32[1,H,N,S]; values u^4 of uniform u (exact multiplications) so rows are peaky."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_layers):
        u = rng.random((1, n_heads, n_tok, n_frames), dtype=np.float32)
        u = u * u
        out.append((u * u).astype(np.float32))
    return out


class _Captured:
    cost = None


def load_reference_align():
    dtw = types.ModuleType("dtw")

    class _SP:
        symmetric1 = "symmetric1"

    class _Res:
        pass

    def _dtw(cost, keep_internals=False, step_pattern=None):
        assert step_pattern == "symmetric1" and cost.dtype == np.float64
        _Captured.cost = np.array(cost, copy=True)
        r = _Res()
        r.index1, r.index2 = O.dtw_symmetric1(cost)
        return r

    dtw.dtw = _dtw
    dtw.stepPattern = _SP
    icu = types.ModuleType("icu")
    icu.Locale = object
    icu.BreakIterator = object
    sys.modules["dtw"] = dtw
    sys.modules["icu"] = icu
    spec = importlib.util.spec_from_file_location("ref_align", "/root/reference/urro_whisper/align.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


class _FakeTokenizer:
    """Enough of the whisper tokenizer for perform_word_alignment to reach the DTW call and beyond;
    every text id decodes to a distinct ' wNNN' word so the host word logic stays trivial."""
    sot, eot, transcribe, translate, task = 50258, 50257, 50359, 50358, 50359
    timestamp_begin, sot_prev, no_timestamps = 50364, 50361, 50363

    def decode(self, ids):
        return "".join(f" w{int(i)}" for i in ids)


def golden_ref_align(ref):
    cases = [("tiny.en", 4, 6, 1), ("tiny.en", 4, 6, 5), ("small", 12, 12, 21), ("large-v3-turbo", 4, 20, 9)]
    heads_json = {}
    for k, v in ref._ALIGNMENT_HEADS.items():
        import base64
        import gzip
        arr = np.frombuffer(gzip.decompress(base64.b85decode(v)), dtype=bool)
        heads_json[k] = {"size": int(arr.size), "flat": np.nonzero(arr)[0].tolist()}
    with open(os.path.join(GOLD, "ref_alignment_heads.json"), "w") as f:
        json.dump(heads_json, f, indent=1)

    # the reference's ICU splitter cannot run (no PyICU): replace it by a one-word-per-token splitter so
    # the numeric front half and the boundary code run unmodified
    def _split(text_tokens, tokenizer, lang="en"):
        return ([f"w{t}" for t in text_tokens], [[t] for t in text_tokens], [[i] for i in range(len(text_tokens))])

    ref.split_tokens_with_icu = _split
    out = {}
    for ci, (key, L, H, N) in enumerate(cases):
        att = synth_attention(100 + ci, L, H, N)
        heads = ref._get_alignment_heads(key, L, H)
        text_ids = list(range(1000, 1000 + N))
        full = [50258, 50259, 50359, 50364] + text_ids
        words = ref.perform_word_alignment(full, text_ids, att, _FakeTokenizer(), heads, L, 1500,
                                           medfilt_width=7, qk_scale=1.0, language="en")
        starts = [w["start"] for w in words]
        ends = [w["end"] for w in words]
        out[f"case{ci}_meta"] = np.array([100 + ci, L, H, N], dtype=np.int64)
        out[f"case{ci}_cost"] = _Captured.cost
        out[f"case{ci}_starts"] = np.array(starts, dtype=np.float64)
        out[f"case{ci}_ends"] = np.array(ends, dtype=np.float64)
        print("ref align case", ci, key, "cost", _Captured.cost.shape, "words", len(words))
    out["case_keys"] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(GOLD, "ref_align_cost.npz"), **out)

    # window planner + merger (align.py:235-387)
    plans = []
    for total_s, ntok in [(30.0, 50), (42.5, 77), (61.0, 3), (95.25, 400), (7.0, 1), (46.0, 0)]:
        total = int(total_s * 16000)
        plans.append({"total_samples": total, "n_tokens": ntok,
                      "windows": [list(map(int, w)) for w in
                                  ref.calculate_sliding_windows_for_alignment(total, list(range(ntok)))]})
    rng = np.random.default_rng(7)
    merges = []
    for trial in range(3):
        info = [(0, 480000, 0, 10), (240000, 720000, 8, 20), (480000, 900000, 18, 30)]
        results = []
        for (s, e, _, _) in info:
            t = s / 16000
            words = []
            for k in range(6):
                st = round(float(t + rng.random() * 4), 3)
                en = round(float(st + rng.random() * 1.5), 3)
                words.append({"text": f"w{trial}_{len(results)}_{k}", "start": st, "end": en})
                t = en
            words.append({"text": "untimed", "start": None, "end": None})
            results.append(words)
        merged = ref.merge_overlapping_alignment_results(results, info)
        merges.append({"window_info": [list(i) for i in info], "window_results": results, "merged": merged})
    with open(os.path.join(GOLD, "ref_align_windows.json"), "w") as f:
        json.dump({"plans": plans, "merges": merges}, f, indent=1)


def golden_hf():
    from transformers import WhisperConfig, WhisperFeatureExtractor, WhisperForConditionalGeneration
    dims = dims_for("micro")
    sd = synthetic_state_dict(dims, seed=0)
    cfg = WhisperConfig(vocab_size=dims.vocab, num_mel_bins=dims.n_mels, encoder_layers=dims.enc_layers,
                        decoder_layers=dims.dec_layers, encoder_attention_heads=dims.n_heads,
                        decoder_attention_heads=dims.n_heads, d_model=dims.d_model, encoder_ffn_dim=dims.d_ffn,
                        decoder_ffn_dim=dims.d_ffn, max_source_positions=1500, max_target_positions=448,
                        activation_function="gelu", attn_implementation="eager", pad_token_id=0,
                        bos_token_id=1, eos_token_id=2, decoder_start_token_id=3)
    hf = WhisperForConditionalGeneration(cfg).eval()
    sd_hf = dict(sd)
    sd_hf["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    missing = hf.load_state_dict(sd_hf, strict=False)
    assert not missing.unexpected_keys, missing
    assert all("proj_out" in k for k in missing.missing_keys), missing

    rng = np.random.default_rng(1234)
    audio = (0.1 * rng.standard_normal(480000)).astype(np.float32)
    # log-mel through HF's torch path, with the reference's padding recipe (pad 480000 then trim 3000)
    fe = WhisperFeatureExtractor(feature_size=80)
    padded = np.concatenate([audio, np.zeros(480000, np.float32)])
    mel_hf = fe._torch_extract_fbank_features(padded[None])[0][:, :3000].astype(np.float32)
    ids = [3, 17, 900, 5, 321, 77]
    with torch.no_grad():
        enc_hf = hf.model.encoder(torch.from_numpy(mel_hf)[None]).last_hidden_state
        o = hf(decoder_input_ids=torch.tensor([ids]), encoder_outputs=(enc_hf,), output_attentions=True)
    np.savez_compressed(
        os.path.join(GOLD, "hf_micro.npz"),
        audio_seed=np.array(1234), ids=np.array(ids),
        mel_sub=mel_hf[:, ::37], mel_max=np.array(mel_hf.max()), mel_mean=np.array(mel_hf.mean(dtype=np.float64)),
        enc_sub=enc_hf[0, ::50].numpy(), enc_mean_abs=np.array(enc_hf.abs().mean().item()),
        logits=o.logits[0].numpy(),
        cross_sub=np.stack([a[0, :, :, ::25].numpy() for a in o.cross_attentions]))
    print("hf golden written; enc mean |x| =", float(enc_hf.abs().mean()))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    golden_ref_align(load_reference_align())
    golden_hf()
