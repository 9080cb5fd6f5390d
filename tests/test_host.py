"""Host-side logic that surrounds the GPU path (no GPU needed)."""
import pytest
import numpy as np

from whisper_amd import align as A
from whisper_amd import config as CFG
from whisper_amd.weights import synthetic_state_dict, tensor_plan


class FakeTokenizer:
    sot = 50258
    vocab = {1: " Hello", 2: " wor", 3: "ld", 4: ",", 5: " how", 6: " are", 7: " you", 8: "?"}

    def decode(self, ids):
        return "".join(self.vocab[int(i)] for i in ids)


def test_words_from_boundaries_groups_tokens_and_keeps_punctuation_untimed():
    ids = [1, 2, 3, 4, 5, 6, 7, 8]
    bnd = [0, 10, 20, 30, 35, 40, 50, 60, 60]
    words = A.words_from_boundaries(ids, bnd, FakeTokenizer(), "en")
    assert [w["word"] for w in words] == ["Hello", "world", ",", "how", "are", "you", "?"]
    w = {x["word"]: x for x in words}
    assert (w["Hello"]["start"], w["Hello"]["end"]) == (0.0, 0.2)
    assert (w["world"]["start"], w["world"]["end"]) == (0.2, 0.6)     # two tokens -> boundary[1] .. boundary[3]
    assert w[","]["start"] is None and w["?"]["end"] is None
    assert (w["you"]["start"], w["you"]["end"]) == (1.0, 1.2)


def test_to_global_times_drops_context_words():
    words = [{"word": "a", "start": 1.0, "end": 1.5}, {"word": "b", "start": 3.0, "end": 2.0}, {"word": ",", "start": None, "end": None}]
    out = A.to_global_times(words, start_sample=160000, context_dur=2.0)
    assert out == [{"text": "b", "start": 11.0, "end": 11.0}, {"text": ",", "start": None, "end": None}]


def test_synthetic_weights_are_deterministic_and_complete():
    dims = CFG.dims_for("micro")
    a, b = synthetic_state_dict(dims, 3), synthetic_state_dict(dims, 3)
    assert list(a) == [n for n, _, _ in tensor_plan(dims)]
    assert all(np.array_equal(a[k].numpy(), b[k].numpy()) for k in a)
    assert len(tensor_plan(CFG.dims_for("small"))) == 5 + 12 * 15 + 2 + 2 + 12 * 24 + 2


def test_n_mels_rule_follows_model_string():
    assert CFG.n_mels_for("large-v3") == 128 and CFG.n_mels_for("large-v3-turbo") == 128
    assert CFG.n_mels_for("large") == 80 and CFG.n_mels_for("turbo") == 80 and CFG.n_mels_for("small") == 80


def test_system_icu_word_breaker():
    """The ctypes binding over the system's libicuuc gives ICU word boundaries (align.py:110-116 uses PyICU for the
    same iterator); offsets are UTF-16 units like PyICU's."""
    from whisper_amd import align as A
    if not A._system_icu():
        pytest.skip("no system ICU library")
    t = "Hello, world! it's 3.5 degrees 日本語のテキスト"
    spans = A._icu_c_word_spans(t, "en")
    assert [t[a:b] for a, b in spans] == ["Hello", ",", " ", "world", "!", " ", "it's", " ", "3.5", " ", "degrees", " ",
                                          "日本語", "の", "テキスト"]
    assert spans[0][0] == 0 and spans[-1][1] == len(t) and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    u = "a 😀 b"                                # one astral character = two UTF-16 units
    assert A._icu_c_word_spans(u, "en")[-1][1] == len(u) + 1


def test_merged_decoder_session_surface_and_cross_kv_key():
    """The merged-decoder surface (what core.py:185-192 probes) and the cross-K/V cache key, on a stand-in engine."""
    import torch
    from whisper_amd.session import DecoderSession

    class Eng:
        dims = CFG.dims_for("micro")
        device = torch.device("cpu")
        projected = 0

        def cross_kv(self, B, enc):
            self.projected += 1

    eng = Eng()
    plain, merged = DecoderSession(eng), DecoderSession(eng, merged=True)
    assert [i.name for i in plain.get_inputs()] == ["input_ids", "encoder_hidden_states"]
    names = [i.name for i in merged.get_inputs()]
    assert names[:2] == ["input_ids", "encoder_hidden_states"] and names[-1] == "use_cache_branch"
    assert names[2:6] == [f"past_key_values.0.{s}.{kv}" for s in ("decoder", "encoder") for kv in ("key", "value")]
    assert len(names) == 3 + 4 * eng.dims.dec_layers
    with pytest.raises(ValueError, match="are missing from input feed"):
        merged.run(["logits"], {"input_ids": np.zeros((1, 2), np.int64), "encoder_hidden_states": np.zeros((1, 1500, 128), np.float32)})
    feeds = {n: np.empty((1, 2, 0, 64), np.float32) for n in names[2:-1]}
    feeds.update({"input_ids": np.zeros((1, 2), np.int64), "encoder_hidden_states": np.zeros((1, 1500, 128), np.float32),
                  "use_cache_branch": np.array([False])})
    bad = dict(feeds)
    bad[names[2]] = np.zeros((1, 2, 3, 64), np.float32)
    with pytest.raises(ValueError, match="empty cache"):
        merged.run(["logits"], bad)
    # the reference hands over a fresh astype() copy of the same states on every step: one projection per content
    enc = np.random.default_rng(0).standard_normal((1, 1500, 128)).astype(np.float32)
    for _ in range(3):
        plain._ensure_cross_kv(enc.astype(np.float32).copy())
    assert eng.projected == 1
    plain._ensure_cross_kv(enc * 1.0001)         # another window's states differ everywhere, so also in the sample
    assert eng.projected == 2
