"""Host side of word alignment: window planning, overlap merging and the token-boundary -> word-timestamp
mapping that sits after the GPU stages (attention rows -> cost matrix -> DTW run in libuw_hip).

Mirrors the interface of urro_whisper/align.py for the parts that stay on the host:
  calculate_sliding_windows_for_alignment   align.py:235-302
  merge_overlapping_alignment_results       align.py:305-387
  split_tokens_into_words                   align.py:94-232   (word breaker injectable; ICU when present)
  words_from_boundaries                     align.py:855-1011 (anchor time is always 0.0 there, see below)
  to_global_times                           align.py:660-677 / 510-530
"""
from __future__ import annotations

import unicodedata
from typing import Callable, Dict, List, Optional, Sequence, Tuple

from .config import AUDIO_TIME_PER_TOKEN, N_AUDIO_CTX, SAMPLE_RATE

ALIGNMENT_WINDOW_SIZE_SAMPLES = 30 * SAMPLE_RATE
ALIGNMENT_OVERLAP_SAMPLES = 15 * SAMPLE_RATE
MIN_ALIGNMENT_WINDOW_SAMPLES = int(5.0 * SAMPLE_RATE)


def calculate_sliding_windows_for_alignment(total_samples: int, transcript_tokens: Sequence[int],
                                            window_size_samples: int = ALIGNMENT_WINDOW_SIZE_SAMPLES,
                                            overlap_samples: int = ALIGNMENT_OVERLAP_SAMPLES,
                                            verbose: bool = False) -> List[Tuple[int, int, int, int]]:
    """(start_sample, end_sample, token_start, token_end) per window: tokens are dealt out in proportion
    to audio time, a window never gets an empty token range, a trailing window under 5 s is dropped."""
    n_tok = len(transcript_tokens)
    hop = window_size_samples - overlap_samples
    windows: List[Tuple[int, int, int, int]] = []
    start = 0
    while start < total_samples:
        end = min(start + window_size_samples, total_samples)
        t0 = int((start / total_samples) * n_tok)
        t1 = min(int((end / total_samples) * n_tok), n_tok)
        if t0 >= n_tok:
            break
        if t1 <= t0:
            t1 = min(t0 + 1, n_tok)
        windows.append((start, end, t0, t1))
        if end >= total_samples or (total_samples - (start + hop)) < MIN_ALIGNMENT_WINDOW_SAMPLES:
            break
        start += hop
    return windows


def merge_overlapping_alignment_results(window_results: Sequence[Sequence[Dict]],
                                        window_info: Sequence[Tuple[int, int, int, int]],
                                        verbose: bool = False) -> List[Dict]:
    """Timed words of all windows, sorted by start; runs of words that overlap the running group end form a
    group; from each group the word closest to the centre of its own window survives."""
    tagged = []
    for idx, (words, (s0, s1, _, _)) in enumerate(zip(window_results, window_info)):
        centre = (s0 / SAMPLE_RATE + s1 / SAMPLE_RATE) / 2
        for w in words:
            if w.get("start") is not None and w.get("end") is not None:
                tagged.append((w, centre))
    if not tagged:
        return []
    tagged.sort(key=lambda wc: wc[0]["start"])   # stable, like list.sort on the reference's dict list
    groups: List[List[Tuple[Dict, float]]] = [[tagged[0]]]
    for item in tagged[1:]:
        group_end = max(w["end"] for w, _ in groups[-1])
        if item[0]["start"] >= group_end:
            groups.append([item])
        else:
            groups[-1].append(item)
    merged = []
    for group in groups:
        best = group[0][0]
        if len(group) > 1:
            best_dist = float("inf")
            for w, centre in group:
                dist = abs((w["start"] + w["end"]) / 2 - centre)
                if dist < best_dist:
                    best_dist, best = dist, w
        merged.append({k: v for k, v in best.items() if not k.startswith("_")})
    return merged


# ---------------------------------------------------------------------------------------------------
def _has_letter(s: str) -> bool:
    return any(unicodedata.category(ch).startswith("L") for ch in s)


_ICU_C = None       # (library, symbol suffix) of the system ICU, False once a lookup failed


def _system_icu():
    """libicuuc of the system through ctypes (its C symbols carry the major version: ubrk_open_70)."""
    global _ICU_C
    if _ICU_C is None:
        _ICU_C = False
        try:
            import ctypes
            import ctypes.util
            import re
            path = ctypes.util.find_library("icuuc")
            if path:
                lib = ctypes.CDLL(path)
                m = re.search(r"\.so\.(\d+)", path)
                for suffix in ([f"_{m.group(1)}"] if m else []) + [""] + [f"_{v}" for v in range(85, 48, -1)]:
                    if hasattr(lib, "ubrk_open" + suffix):
                        fn = getattr(lib, "ubrk_open" + suffix)
                        fn.restype = ctypes.c_void_p
                        fn.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int)]
                        for name in ("ubrk_first", "ubrk_next"):
                            g = getattr(lib, name + suffix)
                            g.restype = ctypes.c_int32
                            g.argtypes = [ctypes.c_void_p]
                        getattr(lib, "ubrk_close" + suffix).argtypes = [ctypes.c_void_p]
                        _ICU_C = (lib, suffix)
                        break
        except OSError:
            pass
    return _ICU_C


def _icu_c_word_spans(text: str, lang: str) -> Optional[List[Tuple[int, int]]]:
    """Word-boundary spans from ubrk_open(UBRK_WORD, lang).  Offsets are UTF-16 code units, exactly what PyICU hands
    the reference (align.py:116, 187-189), which slices the Python string with them."""
    icu_c = _system_icu()
    if not icu_c:
        return None
    import ctypes
    lib, suffix = icu_c
    units = text.encode("utf-16-le")
    buf = ctypes.create_string_buffer(units, len(units) + 2)
    status = ctypes.c_int(0)
    bi = getattr(lib, "ubrk_open" + suffix)(1, lang.encode(), ctypes.cast(buf, ctypes.c_void_p), len(units) // 2, ctypes.byref(status))
    if status.value > 0 or not bi:
        return None
    try:
        nxt = getattr(lib, "ubrk_next" + suffix)
        spans, start = [], getattr(lib, "ubrk_first" + suffix)(bi)
        while True:
            end = nxt(bi)
            if end < 0:           # UBRK_DONE
                break
            spans.append((start, end))
            start = end
        return spans
    finally:
        getattr(lib, "ubrk_close" + suffix)(bi)


def default_word_breaker(text: str, lang: str = "en") -> List[Tuple[int, int]]:
    """(start, end) spans covering `text` from ICU's word BreakIterator, as the reference uses (align.py:110-116):
    PyICU when importable, else the system's libicuuc through ctypes; without any ICU, runs of word characters /
    single other characters."""
    try:
        import icu  # type: ignore
        bi = icu.BreakIterator.createWordInstance(icu.Locale(lang))
        bi.setText(text)
        spans, start = [], bi.first()
        for end in bi:
            spans.append((start, end))
            start = end
        return spans
    except ImportError:
        spans = _icu_c_word_spans(text, lang)
        if spans is not None:
            return spans
        import re
        return [(m.start(), m.end()) for m in re.finditer(r"[\w'’]+|\s+|[^\w\s]", text, flags=re.UNICODE)]


def split_tokens_into_words(text_tokens: Sequence[int], tokenizer, lang: str = "en",
                            breaker: Optional[Callable[[str, str], List[Tuple[int, int]]]] = None):
    """Words (letter-bearing only) with the token ids and token indices that overlap them."""
    breaker = breaker or default_word_breaker
    decoded = tokenizer.decode(list(text_tokens)).strip()
    if not decoded:
        return [], [], []
    spans: List[Tuple[int, int]] = []
    span_owner: List[int] = []
    cursor = 0
    for idx, tok in enumerate(text_tokens):
        piece = tokenizer.decode([tok])
        stripped = piece.lstrip()
        letter = _has_letter(stripped)
        found = decoded.find(piece, cursor)
        length = len(piece)
        if found < 0 or (letter and found > cursor + 10):
            alt = decoded.find(stripped, cursor)
            if alt >= 0 and (found < 0 or abs(alt - cursor) < abs(found - cursor)):
                found, length = alt, len(stripped)
        if found < 0:
            if not letter:
                continue
            found, length = cursor, len(stripped)
        if letter:
            spans.append((found, found + length))
            span_owner.append(idx)
        cursor = found + length
    words, word_tokens, word_indices = [], [], []
    for w0, w1 in breaker(decoded, lang):
        word = decoded[w0:w1].strip()
        if not word or not _has_letter(word):
            continue
        idxs = [span_owner[k] for k, (t0, t1) in enumerate(spans) if t0 < w1 and t1 > w0]
        if idxs:
            words.append(word)
            word_indices.append(idxs)
            word_tokens.append([text_tokens[i] for i in idxs])
    return words, word_tokens, word_indices


def words_from_boundaries(text_tokens: Sequence[int], boundaries: Sequence[int], tokenizer, language: str = "en",
                          sot: Optional[int] = None, breaker=None) -> List[Dict]:
    """Token boundary frames -> [{word, start, end}] with untimed entries for non-letter tokens.
    In the reference the anchor time added to every stamp is always 0.0: its timestamp test
    `id >= timestamp_begin and id < eot` can never hold because eot < timestamp_begin (align.py:877)."""
    n = len(boundaries) - 1
    words, _, word_indices = split_tokens_into_words(text_tokens, tokenizer, language, breaker)
    timed = []
    covered = set()
    for word, idxs in zip(words, word_indices):
        inside = [i for i in idxs if i < n]
        if not inside:
            continue
        lo, hi = min(inside), max(inside) + 1
        start = round(round(int(boundaries[lo]) * AUDIO_TIME_PER_TOKEN, 3) + 0.0, 3)
        end_frame = int(boundaries[hi]) if hi < len(boundaries) else N_AUDIO_CTX
        end = max(start, round(round(end_frame * AUDIO_TIME_PER_TOKEN, 3) + 0.0, 3))
        timed.append({"word": word, "start": start, "end": end, "first": idxs[0]})
        covered.update(idxs)
    out: List[Dict] = []
    it = iter(timed)
    cur = next(it, None)
    sot = sot if sot is not None else getattr(tokenizer, "sot", 1 << 30)
    for idx, tok in enumerate(text_tokens):
        if idx in covered:
            if cur is not None and idx == cur["first"]:
                out.append({"word": cur["word"], "start": cur["start"], "end": cur["end"]})
                cur = next(it, None)
        else:
            piece = tokenizer.decode([tok]).strip()
            if piece and tok < sot:
                out.append({"word": piece, "start": None, "end": None})
    return out


def to_global_times(words: Sequence[Dict], start_sample: int, context_dur: float = 0.0) -> List[Dict]:
    """Segment-relative stamps -> global seconds; words that start inside the prepended audio context are
    dropped, ends are clamped to their start (align.py:660-677)."""
    t0 = start_sample / SAMPLE_RATE
    out = []
    for w in words:
        s, e = w.get("start"), w.get("end")
        gs = ge = None
        if s is not None:
            if s < context_dur:
                continue
            gs = round(max(0, s - context_dur) + t0, 3)
        if e is not None:
            ge = round(max(0, e - context_dur) + t0, 3)
            if gs is not None and ge < gs:
                ge = gs
        if w.get("word"):
            out.append({"text": w["word"], "start": gs, "end": ge})
    return out
