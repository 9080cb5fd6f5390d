"""An independent pin for the word breaker.  The reference segments text with PyICU's word BreakIterator
(urro_whisper/align.py:116,187-189); this package calls the same ICU routine through the system libicuuc, and the
end-to-end golden fixture was made with that very function standing in for PyICU -- so the fixture cannot catch a
mistake in it.  Here its spans are checked against a different implementation of the same standard (UAX #29 default
word boundaries as the `regex` module implements them), on every text the golden fixture holds and on punctuation-heavy
samples."""
import json
import os

import pytest

from whisper_amd.align import _icu_c_word_spans

regex = pytest.importorskip("regex")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_e2e.json")

SAMPLES = [
    "Hello, world! It's 3.5 o'clock -- don't panic; e-mail me at 10:30. naïve café",
    " and so, my fellow Americans: ask not what your country can do for you",
    "co-operate re-enter 1,000.50 U.S.A. x_y z's «quoted» “smart” ‘q’ end…",
    "¿Qué tal? Straße façade Ångström — em–dash; tabs\tand\nnewlines",
    "", " ", "a", "...", "[SPEAKER 1] - hi there -",
]


def _texts(node, out):
    if isinstance(node, dict):
        for key, value in node.items():
            if key == "text" and isinstance(value, str):
                out.append(value)
            else:
                _texts(value, out)
    elif isinstance(node, list):
        for value in node:
            _texts(value, out)


def test_icu_spans_equal_uax29_word_boundaries():
    if _icu_c_word_spans("probe", "en") is None:
        pytest.skip("no system ICU library")
    with open(GOLDEN) as fh:
        texts = []
        _texts(json.load(fh), texts)
    assert len(texts) > 100
    for text in SAMPLES + texts:
        spans = _icu_c_word_spans(text, "en")
        units = text.encode("utf-16-le")
        got = [units[2 * a: 2 * b].decode("utf-16-le") for a, b in spans]
        want = [piece for piece in regex.split(r"(?w)\b", text) if piece]
        assert got == want, text
        assert "".join(got) == text
