"""Host logic of whisper_amd.whisperer against outputs of the reference's own core.py (tests/golden/ref_e2e.json,
made by tools/make_golden_e2e.py: urro_whisper.whisperer imported from the reference with its absent third-party
modules replaced by oracle-backed stand-ins).  Here the same oracle sits behind the engine interface, so every
difference is a difference in control flow: prompt construction, chunk loop and context carry-over, text filter,
boundary-to-word mapping over ICU word breaks, forced-alignment windows and their merge."""
import json
import os

import pytest

from tests.helpers import WordTokenizer, synth_audio
from tests.oracle_engine import OracleEngine
from whisper_amd import align as A
from whisper_amd import whisperer
from whisper_amd.config import alignment_heads, alignment_heads_key, dims_for
from whisper_amd.weights import synthetic_state_dict

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_e2e.json")
CASES = {
    "plain_two_chunks": dict(audio_seed=21, n_samples=16000 * 41, max_tokens=22),
    "user_prompt": dict(audio_seed=22, n_samples=16000 * 12, max_tokens=30, prompt="w900 w901, w902"),
    "hyphen_delimiter": dict(audio_seed=23, n_samples=16000 * 63, max_tokens=24, delimiter=" -"),
    "forced_alignment": dict(audio_seed=24, n_samples=16000 * 47, max_tokens=448,
                             transcript=" ".join(f"w{300 + i}" for i in range(26)) + ", w77."),
    "stream_hyphen": dict(audio_seed=25, n_samples=16000 * 38, max_tokens=20, delimiter=" -", stream=True),
    "stream_plain": dict(audio_seed=26, n_samples=16000 * 33, max_tokens=16, stream=True),
    # diarization: 60 s chunks for a "medium" model string, offset follows the last word (core.py:738-746,775-786,1156-1169)
    "speaker_diarization": dict(model="whisper-medium_timestamped", audio_seed=27, n_samples=16000 * 100, max_tokens=26,
                                delimiter=" [SPEAKER 1]"),
    # user prompt + delimiter with prefix and audio-context carry-over (core.py:874-922,1058-1142)
    "prompt_delimiter_context": dict(audio_seed=28, n_samples=16000 * 70, max_tokens=40, prompt="w700, w701.", delimiter=" >"),
    # model string that is a head-table key of another size: no alignment, untimed words (align.py:81-84,595-597)
    "mask_mismatch": dict(model="medium", audio_seed=29, n_samples=16000 * 35, max_tokens=20),
    # failure isolation (core.py:813-837, 934-980, 1009-1044, 578-597): the golden was made with a session stand-in that
    # raises on the stated runs; `faults` describes the same event as the engine shows it (tests/oracle_engine.py)
    "fail_encoder_chunk2_hyphen": dict(audio_seed=41, n_samples=16000 * 75, max_tokens=22, delimiter=" -", faults=dict(fail_windows=[2])),
    "fail_encoder_chunk2_plain": dict(audio_seed=32, n_samples=16000 * 75, max_tokens=22, faults=dict(fail_windows=[2])),
    "fail_alignment_chunk1": dict(audio_seed=33, n_samples=16000 * 63, max_tokens=24, delimiter=" -", faults=dict(no_align=[1])),
    "fail_decode_chunk2_step5": dict(audio_seed=34, n_samples=16000 * 63, max_tokens=24, delimiter=" >",
                                     faults=dict(decode_stop={2: 4}, no_align=[2])),
    "fail_stream_encoder_chunk2": dict(audio_seed=35, n_samples=16000 * 70, max_tokens=18, delimiter=" -", stream=True,
                                       faults=dict(fail_windows=[2])),
    "fail_forced_window2": dict(audio_seed=36, n_samples=16000 * 47, max_tokens=448,
                                transcript=" ".join(f"w{400 + i}" for i in range(24)) + ".", faults=dict(fail_windows=[2])),
}


def heads_for(case_model, dims):
    """The alignment heads the reference resolves for this model string on these shapes (align.py:572-597)."""
    return alignment_heads(alignment_heads_key(case_model, dims.d_model, "en"), dims.dec_layers, dims.n_heads) or []


@pytest.fixture(scope="module")
def engine():
    gold = json.load(open(GOLD))
    dims = dims_for(gold["model"])
    sd = synthetic_state_dict(dims, seed=gold["weights"]["seed"], std=gold["weights"]["std"])
    engines = {}

    def get(case_model, faults=None):
        key = tuple(heads_for(case_model, dims))
        if faults:      # an engine of its own: the injected faults and the window numbering belong to one run
            return OracleEngine(dims, sd, list(key), max_batch=2, **faults)
        if key not in engines:
            engines[key] = OracleEngine(dims, sd, list(key), max_batch=2)
        return engines[key]
    return gold, get


@pytest.mark.parametrize("name", list(CASES))
def test_whisperer_matches_reference_core(engine, name):
    if not A._system_icu():
        pytest.skip("the golden was made with ICU word breaks; no system ICU here")
    gold, get_engine = engine
    kw = dict(CASES[name])
    audio = synth_audio(kw.pop("audio_seed"), kw.pop("n_samples"))
    model = kw.pop("model", gold["model"])
    faults = kw.pop("faults", None)
    res = whisperer(model=model, audio=audio, language="en", tokenizer=WordTokenizer(), engine=get_engine(model, faults), **kw)
    ref = gold["cases"][name]
    if kw.get("stream"):
        assert list(res) == ref["pieces"]
        return
    assert [int(t) for t in res["tokens"]] == ref["tokens"]
    assert res["text"] == ref["text"]
    assert res["language"] == ref["language"]
    assert len(res["words"]) == len(ref["words"])
    for got, exp in zip(res["words"], ref["words"]):
        assert got["text"] == exp["text"]
        assert got["start"] == exp["start"] and got["end"] == exp["end"], (got, exp)


def test_delimiter_factories_match_reference():
    from whisper_amd import delimiters as D
    gold = json.load(open(GOLD))["delimiters"]
    for name, exp in gold.items():
        f = getattr(D, name)
        assert f.regex == exp["regex"] and f() == exp["default"]
    assert D.HYPHEN(3) == gold["HYPHEN"]["count3"]
    assert D.SPEAKER("(", True) == gold["SPEAKER"]["paren_short"] and D.PERSON("<") == gold["PERSON"]["angle"]
