#!/usr/bin/env python3
"""Generates tests/golden/ref_e2e.json: outputs of the reference's OWN `whisperer` (urro_whisper/core.py, imported
from /root/reference) with every absent third-party module replaced by a stand-in that computes with this
repository's CPU oracle:

  onnxruntime   InferenceSession objects backed by oracle.OracleModel ("tiny" shapes, synthetic weights seed 0),
                exposing input_ids / encoder_hidden_states -> logits, cross_attentions.{l}
  whisper       log_mel_spectrogram = the oracle's log-mel; tokenizer = tests.helpers.WordTokenizer
  dtw           dtw() = oracle.dtw_symmetric1 (the restated dtw-python recursion)
  icu           Locale / BreakIterator over the system's libicuuc (whisper_amd.align's ctypes binding)
  soundfile, bidi.mirror   import-time stand-ins

So the golden pins the reference's control flow (prompt construction, greedy loop, text filter, per-token attention
replay, head selection, boundary extraction, ICU word grouping, chunk loop, context carry-over, forced-alignment
windows) — not third-party numerics.  Runs only in the build container; tests read the committed JSON.
"""
import json
import os
import re
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
GOLD = os.path.join(ROOT, "tests", "golden")

from oracle import whisper_oracle as O  # noqa: E402
from tests.helpers import WordTokenizer, synth_audio  # noqa: E402
from whisper_amd import align as A  # noqa: E402
from whisper_amd.config import dims_for  # noqa: E402
from whisper_amd.weights import synthetic_state_dict  # noqa: E402

MODEL = "tiny"
DIMS = dims_for(MODEL)
SD = synthetic_state_dict(DIMS, seed=0, std=0.05)
OM = O.OracleModel(SD, DIMS.n_heads)


class _IO:
    def __init__(self, name):
        self.name = name


# Fault injection for the failure-isolation cases: which session runs raise.  Chunks (windows) are counted by encoder
# runs, one per chunk; `logits_step` = (chunk, k): the k-th logits run of that chunk raises.
FAULTS = {"encoder_runs": set(), "attn_chunks": set(), "logits_step": None}
STATE = {"encoder_runs": 0, "logits_runs": 0}


def set_faults(encoder_runs=(), attn_chunks=(), logits_step=None):
    FAULTS.update(encoder_runs=set(encoder_runs), attn_chunks=set(attn_chunks), logits_step=logits_step)
    STATE.update(encoder_runs=0, logits_runs=0)


class _Session:
    def __init__(self, path, sess_options=None, providers=None):
        self.kind = "encoder" if "encoder" in os.path.basename(path) else "decoder"
        self.providers = providers or ["CPUExecutionProvider"]

    def get_providers(self):
        return self.providers

    def get_inputs(self):
        return [_IO("input_features")] if self.kind == "encoder" else [_IO("input_ids"), _IO("encoder_hidden_states")]

    def get_outputs(self):
        if self.kind == "encoder":
            return [_IO("last_hidden_state")]
        return [_IO("logits")] + [_IO(f"cross_attentions.{l}") for l in range(DIMS.dec_layers)]

    def run(self, names, feeds):
        if self.kind == "encoder":
            STATE["encoder_runs"] += 1
            STATE["logits_runs"] = 0
            if STATE["encoder_runs"] in FAULTS["encoder_runs"]:
                raise RuntimeError(f"injected encoder failure on run {STATE['encoder_runs']}")
            return [OM.encode(feeds["input_features"])]
        ids = feeds["input_ids"][0].tolist()
        enc = feeds["encoder_hidden_states"]
        names = names or [o.name for o in self.get_outputs()]
        want_attn = any("cross_attentions" in n for n in names)
        chunk = STATE["encoder_runs"]
        if want_attn and chunk in FAULTS["attn_chunks"]:
            raise RuntimeError(f"injected decoder failure (attention outputs) in chunk {chunk}")
        if not want_attn:
            STATE["logits_runs"] += 1
            if FAULTS["logits_step"] == (chunk, STATE["logits_runs"]):
                raise RuntimeError(f"injected decoder failure at step {STATE['logits_runs']} of chunk {chunk}")
        if want_attn:
            logits, cross = OM.decode_full(ids, enc, want_attn=True)
        else:
            logits, cross = OM.decode_full(ids, enc), None
        out = []
        for n in names:
            out.append(logits if n == "logits" else cross[int(re.search(r"\.(\d+)", n).group(1))])
        return out


def install_stubs():
    ort = types.ModuleType("onnxruntime")
    ort.get_available_providers = lambda: ["CPUExecutionProvider"]
    ort.SessionOptions = type("SessionOptions", (), {"log_severity_level": 3})
    ort.InferenceSession = _Session
    sys.modules["onnxruntime"] = ort

    whisper = types.ModuleType("whisper")
    whisper.audio = types.ModuleType("whisper.audio")
    whisper.audio.HOP_LENGTH = 160
    whisper.log_mel_spectrogram = lambda audio, n_mels=80, padding=0: torch.from_numpy(O.log_mel_segment(audio.numpy(), n_mels)[0])
    tok = types.ModuleType("whisper.tokenizer")
    tok.get_tokenizer = lambda multilingual, task=None, language=None: WordTokenizer()
    whisper.tokenizer = tok
    sys.modules.update({"whisper": whisper, "whisper.audio": whisper.audio, "whisper.tokenizer": tok})

    dtw = types.ModuleType("dtw")

    def _dtw(cost, keep_internals=False, step_pattern=None):
        assert step_pattern == "symmetric1" and cost.dtype == np.float64
        r = types.SimpleNamespace()
        r.index1, r.index2 = O.dtw_symmetric1(cost)
        return r

    dtw.dtw = _dtw
    dtw.stepPattern = types.SimpleNamespace(symmetric1="symmetric1")
    sys.modules["dtw"] = dtw

    icu = types.ModuleType("icu")

    class Locale:
        def __init__(self, lang):
            self.lang = lang

    class BreakIterator:
        def __init__(self, lang):
            self.lang, self.bounds = lang, []

        @staticmethod
        def createWordInstance(locale):
            return BreakIterator(locale.lang)

        def setText(self, text):
            spans = A._icu_c_word_spans(text, self.lang)
            assert spans is not None, "system ICU missing"
            self.bounds = [e for _, e in spans]

        def first(self):
            return 0

        def __iter__(self):
            return iter(self.bounds)

    icu.Locale, icu.BreakIterator = Locale, BreakIterator
    sys.modules["icu"] = icu

    sys.modules["soundfile"] = types.ModuleType("soundfile")
    bidi = types.ModuleType("bidi")
    mirror = types.ModuleType("bidi.mirror")
    mirror.MIRRORED = {"[": "]", "]": "[", "(": ")", ")": "(", "<": ">", ">": "<", "{": "}", "}": "{"}
    bidi.mirror = mirror
    sys.modules.update({"bidi": bidi, "bidi.mirror": mirror})


def cases():
    """name -> whisperer kwargs (audio regenerated from seeds by the tests)."""
    return {
        "plain_two_chunks": dict(audio_seed=21, n_samples=16000 * 41, max_tokens=22),
        "user_prompt": dict(audio_seed=22, n_samples=16000 * 12, max_tokens=30, prompt="w900 w901, w902"),
        "hyphen_delimiter": dict(audio_seed=23, n_samples=16000 * 63, max_tokens=24, delimiter=" -"),
        "forced_alignment": dict(audio_seed=24, n_samples=16000 * 47, max_tokens=448,
                                 transcript=" ".join(f"w{300 + i}" for i in range(26)) + ", w77."),
        "stream_hyphen": dict(audio_seed=25, n_samples=16000 * 38, max_tokens=20, delimiter=" -", stream=True),
        "stream_plain": dict(audio_seed=26, n_samples=16000 * 33, max_tokens=16, stream=True),
        # diarization mode (core.py:738-746, 775-786, 1156-1169): a model string containing "medium" selects 60 s chunks
        # (only their first 30 s are encoded) and the next chunk starts at the last word's end when that advances > 10 s;
        # the string is no key of the alignment-head table, so the mask is still picked by width (tiny.en)
        "speaker_diarization": dict(model="whisper-medium_timestamped", audio_seed=27, n_samples=16000 * 100, max_tokens=26,
                                    delimiter=" [SPEAKER 1]"),
        # user prompt + segmentation delimiter: prompt ids in front of every chunk, prefix and audio context carried over
        # (core.py:874-922, 1058-1142)
        "prompt_delimiter_context": dict(audio_seed=28, n_samples=16000 * 70, max_tokens=40, prompt="w700, w701.", delimiter=" >"),
        # a model string that IS a key of the alignment-head table but whose mask has another size than layers x heads:
        # no alignment at all, words come back untimed (align.py:81-84, 595-597; core.py:1041-1052)
        "mask_mismatch": dict(model="medium", audio_seed=29, n_samples=16000 * 35, max_tokens=20),
        # ---- failure isolation (core.py:813-837, 934-980, 1009-1044, 578-597): `faults` goes to set_faults() ----
        # the encoder run of chunk 2 raises: the chunk is skipped, prefix and audio context start over, chunks 1 and 3 stay
        "fail_encoder_chunk2_hyphen": dict(audio_seed=41, n_samples=16000 * 75, max_tokens=22, delimiter=" -",
                                           faults=dict(encoder_runs=[2])),
        "fail_encoder_chunk2_plain": dict(audio_seed=32, n_samples=16000 * 75, max_tokens=22, faults=dict(encoder_runs=[2])),
        # every attention-output run of chunk 1 raises: its words come back untimed and only feed the next prefix
        "fail_alignment_chunk1": dict(audio_seed=33, n_samples=16000 * 63, max_tokens=24, delimiter=" -",
                                      faults=dict(attn_chunks=[1])),
        # the 5th decoder run of chunk 2 raises (and so does its alignment): four tokens are kept, words untimed
        "fail_decode_chunk2_step5": dict(audio_seed=34, n_samples=16000 * 63, max_tokens=24, delimiter=" >",
                                         faults=dict(logits_step=(2, 5), attn_chunks=[2])),
        "fail_stream_encoder_chunk2": dict(audio_seed=35, n_samples=16000 * 70, max_tokens=18, delimiter=" -", stream=True,
                                           faults=dict(encoder_runs=[2])),
        "fail_forced_window2": dict(audio_seed=36, n_samples=16000 * 47, max_tokens=448,
                                    transcript=" ".join(f"w{400 + i}" for i in range(24)) + ".", faults=dict(encoder_runs=[2])),
    }


def main():
    install_stubs()
    sys.path.insert(0, "/root/reference")
    import urro_whisper  # noqa: E402  (the reference package itself)
    import tempfile
    tmp = tempfile.mkdtemp()
    enc_p, dec_p = os.path.join(tmp, "encoder_model.onnx"), os.path.join(tmp, "decoder_model.onnx")
    for p in (enc_p, dec_p):
        open(p, "wb").close()
    out = {"model": MODEL, "weights": {"seed": 0, "std": 0.05}, "cases": {}}
    for name, kw in cases().items():
        kw = dict(kw)
        audio = synth_audio(kw.pop("audio_seed"), kw.pop("n_samples"))
        set_faults(**kw.pop("faults", {}))
        res = urro_whisper.whisperer(model=kw.pop("model", MODEL), audio=audio, language="en", onnx_encoder=enc_p, onnx_decoder=dec_p,
                                     onnx_providers=["CPUExecutionProvider"], **kw)
        if kw.get("stream"):
            pieces = list(res)
            out["cases"][name] = {"pieces": pieces}
            print(name, "pieces", len(pieces))
            continue
        out["cases"][name] = {"text": res["text"], "tokens": [int(t) for t in res["tokens"]], "words": res["words"],
                              "language": res.get("language")}
        timed = sum(1 for w in res["words"] if w.get("start") is not None)
        print(name, "tokens", len(res["tokens"]), "words", len(res["words"]), "timed", timed)
    from urro_whisper import delimiters as RD
    out["delimiters"] = {n: {"regex": getattr(RD, n).regex, "default": getattr(RD, n)()} for n in ("HYPHEN", "GREATER_THAN", "SPEAKER", "PERSON")}
    out["delimiters"]["HYPHEN"]["count3"] = RD.HYPHEN(3)
    out["delimiters"]["SPEAKER"]["paren_short"] = RD.SPEAKER("(", True)
    out["delimiters"]["PERSON"]["angle"] = RD.PERSON("<")
    with open(os.path.join(GOLD, "ref_e2e.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
