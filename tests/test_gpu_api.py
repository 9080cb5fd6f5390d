"""GPU tests of the drop-in boundary: ORT-shaped sessions and whisperer() on top of the HIP engine."""
import re

import numpy as np
import pytest
import torch

from oracle import whisper_oracle as O
from tests.helpers import synth_audio
from whisper_amd import whisperer
from whisper_amd.config import dims_for
from whisper_amd.model import load_onnx_models
from whisper_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu


class MicroTokenizer:
    """Whisper-tokenizer-shaped stand-in for the 1024-entry micro vocabulary: id t < 800 decodes to ' w<t>',
    ids 800..822 to punctuation, specials as in the real layout (eot < sot < ... < timestamps)."""
    eot, sot, sot_prev, no_timestamps, timestamp_begin = 823, 824, 827, 828, 924
    transcribe, translate, task = 826, 829, 826
    language = "en"

    def decode(self, ids):
        out = []
        for t in ids:
            t = int(t)
            if t < 800:
                out.append(f" w{t}")
            elif t < 823:
                out.append(",")
        return "".join(out)

    def encode(self, text, allowed_special=None):
        specials = {"<|en|>": 825, "<|transcribe|>": 826, "<|30.00|>": 924 + 99}
        if text in specials:
            return [specials[text]]
        ids = []
        for m in re.finditer(r"w(\d+)|,|-", text):
            ids.append(int(m.group(1)) if m.group(1) else (801 if m.group(0) == "," else 802))
        return ids


@pytest.fixture(scope="module")
def sessions():
    enc, dec = load_onnx_models("synthetic:micro", None, onnx_providers=["MI355XExecutionProvider"], dtype="f32", max_batch=2)
    yield enc, dec
    enc.engine.close()


def test_sessions_look_like_onnxruntime(sessions):
    enc, dec = sessions
    assert enc.get_providers() == ["MI355XExecutionProvider"]
    assert [i.name for i in dec.get_inputs()] == ["input_ids", "encoder_hidden_states"]
    names = [o.name for o in dec.get_outputs()]
    assert names[0] == "logits" and [n for n in names if "cross_attentions" in n] == ["cross_attentions.0", "cross_attentions.1"]
    with pytest.raises(Exception, match="are missing from input feed"):
        dec.run(["logits"], {"input_ids": np.zeros((1, 2), np.int64)})
    with pytest.raises(RuntimeError):
        load_onnx_models("synthetic:micro", None, onnx_providers=["CUDAExecutionProvider"])


def test_sessions_reproduce_reference_loops(sessions):
    """Drive the sessions exactly as core.py:824-830, 941-968 and align.py:608-618 do."""
    enc, dec = sessions
    dims = dims_for("micro")
    om = O.OracleModel(synthetic_state_dict(dims, 0), dims.n_heads)
    mel = O.log_mel_segment(synth_audio(4), 80)
    hidden = enc.run(None, {enc.get_inputs()[0].name: mel.astype(np.float32)})[0]
    assert hidden.shape == (1, 1500, dims.d_model)
    assert np.abs(hidden - om.encode(mel)).max() < 1e-4
    cur = [824, 825, 826, 924]
    for _ in range(5):
        logits = dec.run(["logits"], {"input_ids": np.array([cur], dtype=np.int64), "encoder_hidden_states": hidden.astype(np.float32)})[0]
        ref = om.decode_full(cur, hidden)
        assert logits.shape == ref.shape and np.abs(logits - ref).max() < 1e-4
        cur.append(int(np.argmax(logits[0, -1, :])))
    attn_names = [o.name for o in dec.get_outputs() if "cross_attentions" in o.name]
    outs = dec.run(attn_names, {"input_ids": np.array([cur], dtype=np.int64), "encoder_hidden_states": hidden.astype(np.float32)})
    _, cross = om.decode_full(cur, hidden, want_attn=True)
    for got, ref in zip(outs, cross):
        assert got.shape == (1, dims.n_heads, len(cur), 1500) and np.abs(got - ref).max() < 1e-6


def test_merged_decoder_surface_runs_the_reference_dummy_cache_branch(sessions):
    """decoder_model_merged.onnx is the file the reference prefers (download.py:259).  Its session declares the cache
    inputs; the reference then feeds EMPTY caches and use_cache_branch=False on every call (core.py:185-192, 840-871,
    943-946; align.py:611-614).  Driven that way the merged surface gives what the plain one gives."""
    import re as _re
    from whisper_amd.session import DecoderSession
    enc, dec = sessions
    dims = dims_for("micro")
    merged = DecoderSession(enc.engine, merged=True)
    names = [i.name for i in merged.get_inputs()]
    kv_names = [n for n in names if n.startswith("past_key_values")]
    assert "use_cache_branch" in names and len(kv_names) == 4 * dims.dec_layers
    n_layers = 1 + max(int(_re.search(r"past_key_values\.(\d+)\.", n).group(1)) for n in kv_names)      # core.py:843-848
    assert n_layers == dims.dec_layers
    hidden = enc.run(None, {enc.get_inputs()[0].name: O.log_mel_segment(synth_audio(4), 80).astype(np.float32)})[0]
    past = {n: np.empty((1, dims.d_model // 64, 0, 64), dtype=np.float32) for n in kv_names}               # core.py:859-862
    cur = [824, 825, 826, 924, 7]
    attn_names = [o.name for o in merged.get_outputs() if "cross_attentions" in o.name]
    for _ in range(3):
        feeds = {"input_ids": np.array([cur], dtype=np.int64), "encoder_hidden_states": hidden.astype(np.float32)}
        plain = dec.run(["logits"], feeds)[0]
        feeds.update(past)
        feeds["use_cache_branch"] = np.array([False], dtype=bool)
        got = merged.run(["logits"], feeds)[0]
        assert np.array_equal(got, plain)
        for a, b in zip(merged.run(attn_names, feeds), dec.run(attn_names, {k: feeds[k] for k in ("input_ids", "encoder_hidden_states")})):
            assert np.array_equal(a, b)
        cur.append(int(np.argmax(got[0, -1])))
    with pytest.raises(Exception, match="are missing from input feed"):
        merged.run(["logits"], {"input_ids": np.array([cur], dtype=np.int64), "encoder_hidden_states": hidden})
    feeds["use_cache_branch"] = np.array([True], dtype=bool)
    with pytest.raises(ValueError, match="use_cache_branch"):
        merged.run(["logits"], feeds)


def _engine(max_batch=4):
    from whisper_amd.engine import Engine
    dims = dims_for("micro")
    return Engine(dims, synthetic_state_dict(dims, 0, std=0.06), dtype="f32", max_batch=max_batch, align_heads=[(0, 1), (1, 0), (1, 1)])


def test_whisperer_result_contract_and_batched_chunks():
    eng = _engine()
    audio = np.concatenate([synth_audio(7), synth_audio(8, 200000)])       # 42.5 s -> two chunks
    res = whisperer("synthetic:micro", audio, language="en", max_tokens=24, tokenizer=MicroTokenizer(), engine=eng)
    assert set(res) == {"text", "words", "tokens", "language"} and res["language"] == "en"
    assert len(res["tokens"]) == 2 * 20 and all(isinstance(t, int) for t in res["tokens"])
    assert res["text"] and all(set(w) == {"text", "start", "end"} for w in res["words"])
    starts = [w["start"] for w in res["words"]]
    # DTW always spans the 1500 encoder frames, so stamps of the short second chunk run up to 30 s + 30 s (as in the reference)
    assert all(0.0 <= s <= 60.0 for s in starts) and any(s >= 30.0 for s in starts)
    assert all(w["end"] >= w["start"] for w in res["words"])
    # chunks are independent without a delimiter: transcribing the first 30 s alone gives the same first 20 tokens
    first = whisperer("synthetic:micro", audio[:480000], max_tokens=24, tokenizer=MicroTokenizer(), engine=eng)
    assert first["tokens"] == res["tokens"][:20]
    with pytest.raises(TypeError):
        whisperer("synthetic:micro", [1, 2, 3], tokenizer=MicroTokenizer(), engine=eng)
    with pytest.raises(ValueError):
        whisperer("synthetic:micro", audio, transcript="w1 w2", stream=True, tokenizer=MicroTokenizer(), engine=eng)
    eng.close()


def test_whisperer_delimiter_stream_and_transcript_modes():
    eng = _engine()
    tok = MicroTokenizer()
    audio = np.concatenate([synth_audio(9), synth_audio(10, 100000)])
    res = whisperer("synthetic:micro", audio, delimiter=" -", max_tokens=30, tokenizer=tok, engine=eng)
    assert len(res["tokens"]) > 0 and isinstance(res["text"], str)
    pieces = list(whisperer("synthetic:micro", audio[:480000], stream=True, max_tokens=16, tokenizer=tok, engine=eng))
    assert pieces and all(p.startswith(" w") or p == "," for p in pieces)
    transcript = " ".join(f"w{i}" for i in range(5, 45))
    al = whisperer("synthetic:micro", audio, transcript=transcript, tokenizer=tok, engine=eng)
    assert al["text"] == transcript and al["tokens"] == list(range(5, 45))
    assert al["words"] and all(w["start"] is not None and 0 <= w["start"] <= len(audio) / 16000 + 30 for w in al["words"])
    assert [w["start"] for w in al["words"]] == sorted(w["start"] for w in al["words"])
    eng.close()


def test_whisperer_on_gpu_reproduces_reference_core_golden():
    """fp32 engine on the golden made from the reference's own core.py (tests/golden/ref_e2e.json): identical tokens
    and text, every word time within one 20 ms frame (measured: equal)."""
    import json
    import os
    from tests.helpers import WordTokenizer
    from tests.oracle_engine import FaultInjector
    from tests.test_reference_e2e import CASES, GOLD, heads_for
    from whisper_amd import engine as E
    gold = json.load(open(GOLD))
    dims = dims_for(gold["model"])
    sd = synthetic_state_dict(dims, seed=gold["weights"]["seed"], std=gold["weights"]["std"])
    engines = {}
    try:
        for name in CASES:
            kw = dict(CASES[name])
            audio = synth_audio(kw.pop("audio_seed"), kw.pop("n_samples"))
            model = kw.pop("model", gold["model"])
            heads = tuple(heads_for(model, dims))      # the mask the reference resolves for this model string ([] on a mismatch)
            if heads not in engines:
                engines[heads] = E.Engine(dims, sd, dtype="f32", max_batch=2, align_heads=list(heads))
            faults = kw.pop("faults", None)     # failure-isolation cases: the same faults, shown by the real engine
            eng = FaultInjector(engines[heads], **faults) if faults else engines[heads]
            res = whisperer(model=model, audio=audio, language="en", tokenizer=WordTokenizer(), engine=eng, **kw)
            ref = gold["cases"][name]
            if kw.get("stream"):      # generator mode (core.py:238-536): the yielded pieces, in order
                assert list(res) == ref["pieces"], name
                continue
            assert [int(t) for t in res["tokens"]] == ref["tokens"], name
            assert res["text"] == ref["text"]
            assert [w["text"] for w in res["words"]] == [w["text"] for w in ref["words"]]
            close = [abs(g["start"] - e["start"]) <= 0.0201 and abs(g["end"] - e["end"]) <= 0.0201
                     for g, e in zip(res["words"], ref["words"]) if e["start"] is not None]
            assert all(close), (name, sum(close), len(close))
    finally:
        for eng in engines.values():
            eng.close()


def test_context_without_alignment_heads_transcribes_untimed():
    """A model key without a mask, or a mask of another size, means "no alignment" in the reference (align.py:81-84,
    595-597): transcription still works, words come back untimed (core.py:1041-1044), streaming is unaffected."""
    from whisper_amd.engine import Engine
    dims = dims_for("micro")
    sd = synthetic_state_dict(dims, 0, std=0.06)
    eng = Engine(dims, sd, dtype="f32", max_batch=2)           # align_heads=None
    ref = _engine(2)
    try:
        tok = MicroTokenizer()
        audio = synth_audio(7)
        res = whisperer("synthetic:micro", audio, max_tokens=24, tokenizer=tok, engine=eng)
        timed = whisperer("synthetic:micro", audio, max_tokens=24, tokenizer=tok, engine=ref)
        assert res["tokens"] == timed["tokens"] and res["text"] == timed["text"] and res["text"]
        assert res["words"] == []                              # only timed words reach result["words"] (core.py:1051-1052)
        out = eng.transcribe_windows(torch.from_numpy(audio[None]).cuda(), torch.tensor([480000], dtype=torch.int32),
                                     [[824, 825, 826, 924]], 24, 823, 824, 924, 1023)
        assert int(out["boundaries"][0, 0]) == -1
        pieces = list(whisperer("synthetic:micro", audio, stream=True, max_tokens=16, tokenizer=tok, engine=eng))
        assert pieces == list(whisperer("synthetic:micro", audio, stream=True, max_tokens=16, tokenizer=tok, engine=ref))
        al = whisperer("synthetic:micro", audio, transcript="w5 w6 w7", tokenizer=tok, engine=eng)
        assert al["tokens"] == [5, 6, 7] and al["words"] == []
    finally:
        eng.close()
        ref.close()


def test_whisperer_through_engine_pool_equals_single_engine():
    """Independent chunks (no delimiter) and forced-alignment windows go through an EnginePool several groups at a time;
    tokens, text and every word time must equal what one engine gives group by group."""
    from whisper_amd.parallel import EnginePool
    from whisper_amd.weights import synthetic_state_dict
    dims = dims_for("micro")
    sd = synthetic_state_dict(dims, 0, std=0.06)
    sel = [(0, 1), (1, 0), (1, 1)]
    tok = MicroTokenizer()
    audio = np.concatenate([synth_audio(20 + i) for i in range(5)] + [synth_audio(30, 150000)])      # six chunks
    transcript = " ".join(f"w{i}" for i in range(5, 125))
    eng = _engine(2)
    try:
        want = whisperer("synthetic:micro", audio, max_tokens=24, tokenizer=tok, engine=eng)
        want_al = whisperer("synthetic:micro", audio, transcript=transcript, tokenizer=tok, engine=eng)
    finally:
        eng.close()
    with EnginePool(dims, sd, inflight=3, dtype="f32", max_batch=2, align_heads=sel) as pool:
        got = whisperer("synthetic:micro", audio, max_tokens=24, tokenizer=tok, engine=pool)
        got_al = whisperer("synthetic:micro", audio, transcript=transcript, tokenizer=tok, engine=pool)
        pieces = list(whisperer("synthetic:micro", audio[:480000], stream=True, max_tokens=16, tokenizer=tok, engine=pool))
    assert got == want and len(got["tokens"]) == 6 * 20
    assert got_al == want_al and got_al["words"]
    assert pieces


@pytest.mark.parametrize("sharp", [1.0, 4.0])
def test_bf16_word_times_within_one_frame_of_fp32(sharp):
    """The contract's bf16 leg for word timestamps, on the teacher-forced path (forced alignment: the same tokens in both
    precisions, so only the attention rows, the cost matrix and the DTW path can differ): every word start and end of
    the bf16 engine lies within one 20 ms frame of the fp32 engine's, whose times equal the CPU oracle's (goldens above).
    `sharp` scales the cross-attention q and k projections: 1 is the flat landscape of random weights, 4 gives peaked
    rows; beyond about 8 the scores are so large that bf16 rounding of q.k moves whole paths (not asserted)."""
    from whisper_amd.engine import Engine
    dims = dims_for("micro")
    sd = synthetic_state_dict(dims, 0, std=0.06)
    for k in list(sd):
        if "encoder_attn.q_proj.weight" in k or "encoder_attn.k_proj.weight" in k:
            sd[k] = sd[k] * sharp
    tok = MicroTokenizer()
    audio = np.concatenate([synth_audio(7), synth_audio(8, 200000)])
    transcript = " ".join(f"w{i}" for i in range(5, 65))
    out = {}
    for dt in ("f32", "bf16"):
        eng = Engine(dims, sd, dtype=dt, max_batch=4, align_heads=[(0, 1), (1, 0), (1, 1)])
        try:
            out[dt] = whisperer("synthetic:micro", audio, transcript=transcript, tokenizer=tok, engine=eng)
        finally:
            eng.close()
    a, b = out["f32"]["words"], out["bf16"]["words"]
    assert len(a) == len(b) and len(a) > 20 and [w["text"] for w in a] == [w["text"] for w in b]
    for x, y in zip(a, b):
        assert abs(x["start"] - y["start"]) <= 0.0201 and abs(x["end"] - y["end"]) <= 0.0201, (x, y)


def test_onnx_file_pair_runs_on_the_gpu(tmp_path):
    """The reference's own file contract (encoder_model.onnx + decoder_model_merged.onnx, download.py:217-276) end to end
    on the GPU: weights read from the files give the same sessions and the same whisperer() result as the state dict they
    were written from.  The files come from the suite's writer (no exported Whisper graph exists offline), so this pins
    the plumbing from file to device, not the reader against a real export."""
    from tests.onnx_writer import write_model
    from whisper_amd.engine import Engine
    dims = dims_for("micro")
    sd = synthetic_state_dict(dims, 0, std=0.06)
    enc_w = {k: v.numpy() for k, v in sd.items() if k.startswith("model.encoder.")}
    dec_w = {k: v.numpy() for k, v in sd.items() if k.startswith("model.decoder.")}
    d = tmp_path / "micro"
    d.mkdir()
    ep, dp = str(d / "encoder_model.onnx"), str(d / "decoder_model_merged.onnx")
    write_model(ep, enc_w, "encoder", "torch", True, False)
    write_model(dp, dec_w, "decoder", "torch", True, True)
    tok = MicroTokenizer()
    audio = np.concatenate([synth_audio(7), synth_audio(8, 200000)])
    ref_eng = Engine(dims, sd, dtype="f32", max_batch=2, align_heads=[])
    try:
        want = whisperer("synthetic:micro", audio, max_tokens=24, tokenizer=tok, engine=ref_eng)
    finally:
        ref_eng.close()
    got = whisperer("micro", audio, max_tokens=24, tokenizer=tok, onnx_encoder=ep, onnx_decoder=dp, dtype="f32", max_batch=2)
    assert got["tokens"] == want["tokens"] and got["text"] == want["text"] and len(got["tokens"]) == 40
    enc_s, dec_s = load_onnx_models(ep, dp, onnx_providers=["MI355XExecutionProvider"], model="micro", dtype="f32", max_batch=1)
    try:
        assert "use_cache_branch" in [i.name for i in dec_s.get_inputs()]       # merged file name -> merged surface
        mel = O.log_mel_segment(synth_audio(4), 80).astype(np.float32)
        hidden = enc_s.run(None, {enc_s.get_inputs()[0].name: mel})[0]
        om = O.OracleModel(sd, dims.n_heads)
        assert np.abs(hidden - om.encode(mel)).max() < 3e-4
    finally:
        enc_s.engine.close()


def test_cached_model_files_are_found_without_paths(tmp_path, monkeypatch):
    """No onnx_encoder / onnx_decoder given: the files the reference keeps under ~/.cache/urro_whisper/models/<name>/ are
    used (download.py:240-259, merged decoder preferred); nothing is downloaded."""
    from tests.onnx_writer import write_model
    from whisper_amd.engine import Engine
    monkeypatch.setenv("HOME", str(tmp_path))
    dims = dims_for("micro")
    sd = synthetic_state_dict(dims, 0, std=0.06)
    d = tmp_path / ".cache" / "urro_whisper" / "models" / "micro"
    d.mkdir(parents=True)
    write_model(str(d / "encoder_model.onnx"), {k: v.numpy() for k, v in sd.items() if k.startswith("model.encoder.")}, "encoder", "torch", True, False)
    write_model(str(d / "decoder_model_merged.onnx"), {k: v.numpy() for k, v in sd.items() if k.startswith("model.decoder.")}, "decoder", "torch", True, True)
    tok = MicroTokenizer()
    audio = synth_audio(9)
    ref_eng = Engine(dims, sd, dtype="f32", max_batch=1, align_heads=[])
    try:
        want = whisperer("synthetic:micro", audio, max_tokens=20, tokenizer=tok, engine=ref_eng)
    finally:
        ref_eng.close()
    got = whisperer("micro", audio, max_tokens=20, tokenizer=tok, dtype="f32", max_batch=1)
    assert got["tokens"] == want["tokens"] and got["text"] == want["text"]
    with pytest.raises(FileNotFoundError, match="not found in cache"):
        whisperer("tiny", audio, max_tokens=20, tokenizer=tok)
