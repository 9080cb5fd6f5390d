"""`whisperer(...)`: the reference's public entry point (urro_whisper/core.py:66-80) over the MI355X engine.

Same arguments, result dictionary, error behaviour and chunk semantics as the reference; the per-chunk numerics
(log-mel, encoder, greedy decode, alignment rows, cost matrix, DTW) run in libuw_hip through
`Engine.transcribe_windows`.  Chunks of one audio depend on each other only when a delimiter is set (prefix and
audio context come from the previous chunk, core.py:1058-1146); without a delimiter they are independent and are
transcribed as one batch.  Extra keyword-only arguments: `tokenizer` (an openai-whisper style tokenizer; when
omitted openai-whisper's is loaded as the reference does), `engine` (reuse a loaded Engine or EnginePool), `dtype`,
`max_batch`, `inflight` (batches of independent windows kept in flight on the GPU; > 1 builds an EnginePool).
"""
from __future__ import annotations

import logging
import math
import re
import unicodedata
from typing import Dict, Generator, List, Optional, Sequence, Union

import numpy as np
import torch

from . import align as A
from .config import CHUNK_LENGTH_SAMPLES, SAMPLE_RATE, alignment_heads, alignment_heads_key, dims_for, n_mels_for
from .delimiters import GREATER_THAN, HYPHEN, PERSON, SPEAKER
from .model.load import get_tokenizer_for_model, resolve_weights, select_providers

logger = logging.getLogger("urro_whisper")
if not logger.hasHandlers():
    _h = logging.StreamHandler()
    _h.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
    logger.addHandler(_h)
logger.setLevel(logging.WARNING)

CONTEXT_WORDS = 5
MAX_CONTEXT_SEARCH_WORDS = 15
MIN_CHUNK_SAMPLES = int(0.2 * SAMPLE_RATE)
SEGMENTATION_DELIMITER_PATTERNS = [HYPHEN.regex, GREATER_THAN.regex]
DIARIZATION_DELIMITER_PATTERNS = [SPEAKER.regex, PERSON.regex, r"(\s)?\[?S\d\]?\s", r"(\s)?\[?P\d\]?\s"]


def _punctuation_of(text: str) -> str:
    return "".join(sorted({ch for ch in text if unicodedata.category(ch).startswith("P")}))


def _is_diarization(delimiter: Optional[str]) -> bool:
    if not (delimiter and isinstance(delimiter, str)):
        return False
    if any(re.search(p, delimiter) for p in DIARIZATION_DELIMITER_PATTERNS):
        return True
    return any(k in delimiter.lower() for k in ("speaker", "person"))


def _load_audio(audio) -> np.ndarray:
    if isinstance(audio, str):
        try:
            import soundfile as sf
        except ImportError as e:
            raise ImportError(f"reading audio files needs soundfile: {e}")
        data, sr = sf.read(audio, dtype="float32")
        if data.ndim > 1:
            data = data.mean(axis=1)
        if sr != SAMPLE_RATE:
            try:
                import librosa
            except ImportError:
                raise ImportError("librosa is required for resampling but not installed")
            data = librosa.resample(y=data, orig_sr=sr, target_sr=SAMPLE_RATE)
        return data.astype(np.float32)
    if isinstance(audio, np.ndarray):
        data = audio.astype(np.float32)
        if audio.ndim > 1:
            logger.warning("input numpy array has >1 dimension, averaging channels")
            data = data.mean(axis=-1)
        return data
    raise TypeError("input 'audio' must be a file path (str) or a numpy array")


def _next_prefix(chunk_words: Sequence[Dict], delimiter: str, diarization: bool) -> str:
    """Delimiter last seen in the chunk + its last five words, the final one stripped of punctuation
    (core.py:1058-1100)."""
    parts = [w["text"] for w in chunk_words if w.get("text")][-CONTEXT_WORDS:]
    current = delimiter
    patterns = SEGMENTATION_DELIMITER_PATTERNS + (DIARIZATION_DELIMITER_PATTERNS if diarization else [])
    limit = min(len(chunk_words), MAX_CONTEXT_SEARCH_WORDS * 2)
    found = False
    for w in reversed(list(chunk_words)[-limit:]):
        text = w.get("text", "").strip()
        for p in patterns:
            if re.fullmatch(p.strip(), text):
                current, found = text + " ", True
                break
        if found:
            break
    if parts:
        clean = parts[-1].strip(_punctuation_of(parts[-1]))
        if clean:
            parts[-1] = clean
        elif len(parts) > 1:
            parts.pop()
    joined = " ".join(parts)
    prefix = current + joined + (" " if joined else "")
    if delimiter.startswith(" "):
        prefix = " " + prefix.lstrip()
    return prefix if prefix.strip() else delimiter


class _Run:
    """Everything one whisperer() call needs: engine, tokenizer, special ids, prompt builder."""

    def __init__(self, model, language, prompt, max_tokens, tokenizer, engine, dtype, max_batch, onnx_encoder, onnx_decoder, inflight=1):
        self.model, self.language, self.max_tokens = model, language, max_tokens
        if tokenizer is None:
            tokenizer, self.multilingual = get_tokenizer_for_model(model, language, "transcribe")
        else:
            self.multilingual = not model.endswith(".en")
        self.tok = tokenizer
        try:
            self.eot, self.sot, self.sot_prev = tokenizer.eot, tokenizer.sot, tokenizer.sot_prev
            self.no_timestamps, self.ts_begin = tokenizer.no_timestamps, tokenizer.timestamp_begin
        except AttributeError as e:
            raise RuntimeError(f"tokenizer missing required tokens: {e}")
        try:
            self.ts_end = tokenizer.encode("<|30.00|>", allowed_special="all")[0]
        except Exception as e:
            raise RuntimeError(f"could not encode timestamp end token: {e}")
        self.owns_engine = engine is None     # an engine built here is closed when the call ends (it holds GBs of HBM)
        if engine is None:
            from .engine import Engine
            name, sd = resolve_weights(model, onnx_encoder, onnx_decoder)
            dims = dims_for(name)
            heads = alignment_heads(alignment_heads_key(name, dims.d_model, language), dims.dec_layers, dims.n_heads) or []
            if inflight > 1:
                from .parallel import EnginePool
                engine = EnginePool(dims, sd, inflight=inflight, dtype=dtype, max_batch=max_batch, align_heads=heads)
            else:
                engine = Engine(dims, sd, dtype=dtype, max_batch=max_batch, align_heads=heads)
        self.eng = engine
        if n_mels_for(model) != engine.dims.n_mels:
            raise ValueError(f"mel spectrogram final shape validation failed: model string '{model}' implies "
                             f"{n_mels_for(model)} mel bins but the weights take {engine.dims.n_mels}")
        self.user_prompt: List[int] = []
        if prompt and isinstance(prompt, str):
            try:
                enc = tokenizer.encode(prompt.strip(), allowed_special="all")
            except Exception as e:
                raise RuntimeError(f"prompt encoding failed: {e}")
            if enc:
                self.user_prompt = [self.sot_prev] + enc
        self.include_timestamps = self.no_timestamps is None or self.no_timestamps not in self.user_prompt

    def close(self) -> None:
        if self.owns_engine and self.eng is not None:
            self.eng.close()
            self.eng = None

    def standard_prompt(self) -> List[int]:
        ids = [self.sot]
        try:
            if self.multilingual:
                ids.append(self.tok.encode(f"<|{self.language}|>", allowed_special="all")[0])
            ids.append(self.tok.encode("<|transcribe|>", allowed_special="all")[0])
        except Exception as e:
            raise RuntimeError(f"language/task token encoding failed: {e}")
        ids.append(self.ts_begin if self.include_timestamps or self.no_timestamps is None else self.no_timestamps)
        return ids

    def chunk_prompt(self, prefix: Optional[str]) -> List[int]:
        ids = self.user_prompt + self.standard_prompt()
        if prefix:
            try:
                ids += list(self.tok.encode(prefix, allowed_special="all"))
            except Exception as e:
                raise RuntimeError(f"prefix encoding failed: {e}")
        return ids

    def is_text(self, t: int) -> bool:
        return t < self.sot and not (self.ts_begin <= t <= self.ts_end)

    # one batch of independent segments -> per segment (new_tokens, text_ids, boundaries, decode_failed), or None
    def transcribe(self, segments: Sequence[np.ndarray], prompts: Sequence[List[int]], first_chunk: int = 1):
        """Failure isolation as in the reference's chunk loop: a chunk whose numerics fail is logged and skipped
        (mel / encoder stage, core.py:813-837) -- its entry is None -- and the chunks before and after it are kept.  A
        group of windows that fails as a whole is retried window by window, so one bad window costs only itself.  A
        window whose decoder produced a non-finite logits row comes back with the tokens generated before that step
        and decode_failed set (the reference's generation-loop failure, core.py:934-980: error stored, partial tokens
        kept)."""
        eng = self.eng
        submit = getattr(eng, "transcribe_windows_async", None)   # an EnginePool keeps several groups in flight

        def call_args(segs, prs):
            stride = max(len(s) for s in segs)
            buf = np.zeros((len(segs), stride), dtype=np.float32)
            for i, s in enumerate(segs):
                buf[i, :len(s)] = s
            n = torch.tensor([len(s) for s in segs], dtype=torch.int32)
            return (torch.from_numpy(buf).to(eng.device), n, list(prs), self.max_tokens, self.eot, self.sot, self.ts_begin, self.ts_end)

        def unpack(res, count, chunk0):
            n_new, n_text = res["n_new"].cpu().numpy(), res["n_text"].cpu().numpy()
            new, text, bnd = res["new_tokens"].cpu().numpy(), res["text_ids"].cpu().numpy(), res["boundaries"].cpu().numpy()
            status = res.get("status")
            status = np.zeros(count, dtype=np.int32) if status is None else np.asarray(status.cpu() if hasattr(status, "cpu") else status)
            out = []
            for i in range(count):
                if status[i]:
                    logger.error(f"error during decoder generation loop for chunk {chunk0 + i}: non-finite logits after "
                                 f"{int(n_new[i])} generated tokens")
                out.append((new[i, :n_new[i]].tolist(), text[i, :n_text[i]].tolist(), bnd[i, :n_text[i] + 1].tolist(), bool(status[i])))
            return out

        pending = []
        for i0 in range(0, len(segments), eng.max_batch):
            segs, prs = segments[i0:i0 + eng.max_batch], prompts[i0:i0 + eng.max_batch]
            try:
                args = call_args(segs, prs)
                pending.append((i0, segs, prs, submit(*args) if submit else eng.transcribe_windows(*args), None))
            except Exception as e:                      # noqa: BLE001 -- the reference catches broadly and skips the chunk
                pending.append((i0, segs, prs, None, e))
        out = []
        for i0, segs, prs, res, err in pending:
            if err is None:
                try:
                    if submit:
                        res = res.result()
                    out += unpack(res, len(segs), first_chunk + i0)
                    continue
                except Exception as e:                  # noqa: BLE001
                    err = e
            if len(segs) == 1:
                logger.error(f"chunk {first_chunk + i0} failed on the engine and is skipped: {err}")
                out.append(None)
                continue
            logger.error(f"a group of {len(segs)} chunks starting at chunk {first_chunk + i0} failed ({err}); retrying them one by one")
            for i, (seg, pr) in enumerate(zip(segs, prs)):
                try:
                    out += unpack(eng.transcribe_windows(*call_args([seg], [pr])), 1, first_chunk + i0 + i)
                except Exception as e:                  # noqa: BLE001
                    logger.error(f"chunk {first_chunk + i0 + i} failed on the engine and is skipped: {e}")
                    out.append(None)
        return out

    def words_for(self, text_ids, boundaries, start_sample, context_dur):
        raw = A.words_from_boundaries(text_ids, boundaries, self.tok, self.language, sot=self.sot)
        return A.to_global_times(raw, start_sample, context_dur)


def whisperer(model: str, audio: Union[str, np.ndarray], language: str = "en", transcript: Optional[str] = None,
              delimiter: Optional[str] = None, prompt: Optional[str] = None, onnx_encoder: Optional[str] = None,
              onnx_decoder: Optional[str] = None, verbose: bool = False, stream: bool = False, max_tokens: int = 448,
              onnx_providers: Optional[List[str]] = None, exclude_providers_on_error: List[str] = ['CoreMLExecutionProvider'],
              *, tokenizer=None, engine=None, dtype: str = "bf16", max_batch: int = 16, inflight: int = 1
              ) -> Union[Dict, Generator[str, None, None]]:
    logger.setLevel(logging.INFO if verbose else logging.WARNING)
    diarization = _is_diarization(delimiter)
    audio_data = _load_audio(audio)
    total = len(audio_data)
    select_providers(onnx_providers, exclude_providers_on_error)
    if transcript is not None and stream:
        raise ValueError("cannot use 'transcript' parameter when 'stream' is true.")
    if transcript is not None:
        prompt, delimiter = None, None
    run = _Run(model, language, prompt, max_tokens, tokenizer, engine, dtype, max_batch, onnx_encoder, onnx_decoder, inflight)
    big_model = "medium" in model.lower() or "large" in model.lower()
    if diarization and not big_model:
        logger.warning("speaker diarization with `speaker()` and `person()` is only supported by model sizes medium, "
                       "large-v1, large-v2, and large-v3")
    if stream:
        return _stream(run, audio_data, delimiter, diarization)     # the generator closes the run when it ends
    try:
        if transcript is not None:
            return _align_transcript(run, audio_data, transcript)
        return _transcribe(run, audio_data, delimiter, diarization, big_model)
    finally:
        run.close()


def _align_transcript(run: _Run, audio: np.ndarray, transcript: str) -> Dict:
    """Sliding-window forced alignment (core.py:542-721): teacher-forced attention rows, no greedy loop."""
    try:
        tokens = list(run.tok.encode(transcript, allowed_special="all"))
    except Exception as e:
        raise RuntimeError(f"transcript encoding failed: {e}")
    windows = A.calculate_sliding_windows_for_alignment(len(audio), tokens)
    eng, prompt = run.eng, run.standard_prompt()[:-1] + [run.ts_begin]
    def boundaries_of(e, buf, n, prompts, texts_live):
        """Teacher-forced rows and DTW for one group of windows, all on one context."""
        mel = e.mel(buf.to(e.device), n)
        e.encode(mel)
        e.cross_kv(len(prompts))
        e.attn_rows_forced(prompts, texts_live)
        return e.align(torch.tensor([len(t) for t in texts_live], dtype=torch.int32)).cpu().numpy()

    submit = getattr(eng, "submit", None)      # an EnginePool keeps several groups in flight

    def group_args(group, texts, live):
        stride = max(group[k][1] - group[k][0] for k in live)
        buf = np.zeros((len(live), stride), dtype=np.float32)
        for r, k in enumerate(live):
            seg = audio[group[k][0]:group[k][1]]
            buf[r, :len(seg)] = seg
        n = torch.tensor([group[k][1] - group[k][0] for k in live], dtype=torch.int32)
        return (torch.from_numpy(buf), n, [prompt] * len(live), [texts[k] for k in live])

    pending = []
    for i0 in range(0, len(windows), eng.max_batch):
        group = windows[i0:i0 + eng.max_batch]
        texts = [[t for t in tokens[w[2]:w[3]] if run.is_text(t)] for w in group]
        # no alignment heads (unknown model key / mask of another size): every window comes back empty (align.py:595-597)
        live = [k for k, t in enumerate(texts) if t and eng.heads and len(prompt) + len(t) - 1 <= eng.n_text_ctx]
        bnd, err = None, None
        if live:
            try:
                args = group_args(group, texts, live)
                bnd = submit(boundaries_of, *args) if submit else boundaries_of(eng, *args)
            except Exception as e:                      # noqa: BLE001 -- a failing window yields no words (core.py:578-597,691-693)
                err = e
        pending.append((i0, group, texts, live, bnd, err))
    results: List[List[Dict]] = []
    for i0, group, texts, live, bnd, err in pending:
        group_words: List[List[Dict]] = [[] for _ in group]
        rows = {}                                       # window index in the group -> its boundary row
        if live and err is None:
            try:
                if submit:
                    bnd = bnd.result()
                rows = {k: bnd[r] for r, k in enumerate(live)}
            except Exception as e:                      # noqa: BLE001
                err = e
        if live and err is not None:
            logger.error(f"alignment of windows {i0 + 1}..{i0 + len(group)} failed as a group ({err}); retrying them one by one")
            for k in live:
                try:
                    one_args = group_args(group, texts, [k])
                    one = submit(boundaries_of, *one_args).result() if submit else boundaries_of(eng, *one_args)
                    rows[k] = one[0]
                except Exception as e:                  # noqa: BLE001
                    logger.error(f"window {i0 + k + 1}: alignment failed: {e}")
        for k, row in rows.items():
            if row[0] >= 0:     # -1: DTW found no path for this window (align.py:823-827)
                group_words[k] = run.words_for(texts[k], row[:len(texts[k]) + 1].tolist(), group[k][0], 0.0)
        results += group_words
    words = A.merge_overlapping_alignment_results(results, windows)
    return {"text": transcript, "words": words, "tokens": tokens, "language": run.language}


def _chunks(total: int, chunk_size: int):
    offset, idx = 0, 0
    while offset < total:
        idx += 1
        end = min(offset + chunk_size, total)
        if (end - offset) < MIN_CHUNK_SAMPLES and idx > 1:
            break
        yield offset, end
        offset = end


def _stream(run: _Run, audio: np.ndarray, delimiter, diarization) -> Generator[str, None, None]:
    """Generator mode (core.py:238-536): decoded text per generated text token, text-only prefix carry-over."""
    def gen():
        try:
            yield from chunks()
        finally:
            run.close()

    def chunks():
        prefix, idx = delimiter, 0
        for start, end in _chunks(len(audio), CHUNK_LENGTH_SAMPLES):
            seg = audio[start:end]
            if len(seg) < MIN_CHUNK_SAMPLES:
                prefix = delimiter
                idx += 1
                continue
            idx += 1
            got = run.transcribe([seg], [run.chunk_prompt(prefix)], first_chunk=idx)[0]
            if got is None:                      # mel / encoder stage failed: skip the chunk, reset the prefix (core.py:286-312)
                prefix = delimiter
                continue
            new, text = got[0], got[1]
            for t in new:
                if run.is_text(t):
                    piece = run.tok.decode([t])
                    if piece and not piece.startswith("�"):
                        yield piece
            words = [{"text": w} for w in run.tok.decode(text).strip().split()] if text else []
            prefix = _next_prefix(words, delimiter, diarization) if (words and delimiter) else delimiter
    return gen()


def _transcribe(run: _Run, audio: np.ndarray, delimiter, diarization, big_model) -> Dict:
    total = len(audio)
    all_words: List[Dict] = []
    texts: List[str] = []
    all_tokens: List[int] = []
    chunk_size = min(60 * SAMPLE_RATE, total) if (diarization and big_model) else CHUNK_LENGTH_SAMPLES

    def finish(new, text, bnd, start, context_dur):
        all_tokens.extend(new)
        raw = run.tok.decode(text).strip() if text else ""
        if raw:
            texts.append(raw)
        # boundaries starting with -1: no alignment heads in the context, or DTW found no path (align.py:595-597,823-827)
        aligned = bool(text) and bool(run.eng.heads) and run.include_timestamps and len(bnd) > 0 and bnd[0] >= 0
        words = run.words_for(text, bnd, start, context_dur) if aligned else []
        if not words and raw:
            words = [{"text": w, "start": None, "end": None} for w in raw.split()]
        all_words.extend(w for w in words if w.get("start") is not None)
        return words

    if not delimiter:
        # independent chunks: one batch (core.py:1149-1153 resets prefix and context after every chunk)
        spans = [(s, e) for s, e in _chunks(total, chunk_size) if e - s >= MIN_CHUNK_SAMPLES]
        prompt_ids = run.chunk_prompt(None)
        outs = run.transcribe([audio[s:e] for s, e in spans], [prompt_ids] * len(spans)) if spans else []
        for (s, e), got in zip(spans, outs):
            if got is not None:                  # a failed chunk contributes nothing (core.py:813-837)
                finish(got[0], got[1], got[2], s, 0.0)
    else:
        offset, idx = 0, 0
        prefix, context = delimiter, np.zeros(0, np.float32)
        while offset < total:
            idx += 1
            start = offset
            if diarization and idx > 1 and all_words and all(w.get("end") is not None for w in all_words[-5:]):
                last_end = int(all_words[-1]["end"] * SAMPLE_RATE)
                if last_end > start:
                    start = last_end
            end = min(start + chunk_size, total)
            if (end - start) < MIN_CHUNK_SAMPLES and idx > 1:
                break
            seg = np.concatenate([context, audio[start:end]])
            context_dur = len(context) / SAMPLE_RATE
            if len(seg) < MIN_CHUNK_SAMPLES:
                offset, context, prefix = end, np.zeros(0, np.float32), delimiter
                continue
            got = run.transcribe([seg], [run.chunk_prompt(prefix)], first_chunk=idx)[0]
            if got is None:
                # mel / encoder stage failed: the chunk is skipped, context and prefix start over (core.py:813-837)
                offset, context, prefix = end, np.zeros(0, np.float32), delimiter
                continue
            words = finish(got[0], got[1], got[2], start, context_dur)
            context = np.zeros(0, np.float32)
            if words:
                prefix = _next_prefix(words, delimiter, diarization)
                if not diarization:
                    timed = [w for w in words[-min(len(words), MAX_CONTEXT_SEARCH_WORDS):]
                             if w.get("start") is not None and w.get("end") is not None]
                    if timed:
                        first = timed[max(0, len(timed) - CONTEXT_WORDS)]
                        t0, t1 = float(first["start"]), max(float(first["start"]), float(timed[-1]["end"]))
                        s0 = max(0, math.floor(t0 * SAMPLE_RATE))
                        s1 = min(total, math.ceil(t1 * SAMPLE_RATE))
                        if s1 > s0:
                            context = audio[s0:s1][-int(5.0 * SAMPLE_RATE):]
            else:
                prefix = delimiter
            if (diarization and words and all_words and all_words[-1].get("end") is not None
                    and int(all_words[-1]["end"] * SAMPLE_RATE) > offset + CHUNK_LENGTH_SAMPLES // 3):
                offset = int(all_words[-1]["end"] * SAMPLE_RATE)
            else:
                offset = end
    return {"text": " ".join(texts).strip(), "words": all_words, "tokens": all_tokens, "language": run.language}
