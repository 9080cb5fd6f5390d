"""CPU restatement of the urro_whisper transcription hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (whisper_amd/) never does and fails loudly when its HIP library is missing.

PARITY STATUS: *parity unpinned* against the reference's own tests -- the reference ships none
(no tests/, no fixtures, no golden vectors) and its numerics live in third-party packages that are
not installed here (onnxruntime 1.21.0, openai-whisper 20240930, dtw-python 1.5.3, PyICU).  What
pins this file instead:
  * reference code that does run here: urro_whisper/align.py loaded with stub `dtw`/`icu` modules
    (tools/make_golden.py) -> alignment-head tables, window planner, overlap merger and the exact
    f64 cost matrix handed to dtw.dtw  (tests/golden/ref_align_*.npz);
  * scipy.ndimage.median_filter and torch softmax/mean, which are the very routines the reference
    calls (align.py:776,801,807);
  * HuggingFace transformers' WhisperModel (the architecture the reference's ONNX graphs are
    exported from) for encoder/decoder math, and its feature extractor for the log-mel formula;
  * dtw-python's published algorithm for `symmetric1`, restated in `dtw_symmetric1` below and
    cross-checked against a brute-force path enumeration on small matrices.

Each function cites the reference lines it follows.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SAMPLE_RATE = 16000
N_FFT = 400
HOP = 160
N_FRAMES = 3000
N_AUDIO_CTX = 1500


# --------------------------------------------------------------------------------------------
# log-mel front end: urro_whisper/audio/load.py:67-121 -> whisper.log_mel_spectrogram (call :83)
# --------------------------------------------------------------------------------------------
def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    with np.errstate(divide="ignore"):
        log_t = min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep
    return np.where(f >= min_log_hz, log_t, mels)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = m * f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(n_mels: int) -> np.ndarray:
    """[n_mels, 201] fp32 triangular Slaney filters, area-normalised: the construction
    openai-whisper's mel_filters.npz was produced with (librosa.filters.mel(sr=16000, n_fft=400))."""
    fftfreqs = np.linspace(0.0, SAMPLE_RATE / 2, N_FFT // 2 + 1)
    mel_pts = np.linspace(_hz_to_mel_slaney(0.0), _hz_to_mel_slaney(SAMPLE_RATE / 2), n_mels + 2)
    mel_f = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, fftfreqs.size), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    w *= enorm[:, None]
    return w.astype(np.float32)


def log_mel_segment(audio: np.ndarray, n_mels: int = 80, dtype=torch.float32) -> np.ndarray:
    """audio/load.py:83-85: log_mel_spectrogram(audio, n_mels, padding=3000*160)[:, :3000], batch dim
    added (:96).  Zero-pad 480000 samples at the end, centred reflect-padded STFT (n_fft 400, hop 160,
    periodic Hann), drop the last frame, |X|^2, mel matmul, log10(clamp 1e-10), floor at the GLOBAL
    max - 8 over every frame (also those past 3000), (x + 4) / 4, then keep 3000 frames."""
    x = torch.from_numpy(np.asarray(audio, dtype=np.float32)).to(dtype)
    x = F.pad(x, (0, N_FRAMES * HOP))
    window = torch.hann_window(N_FFT, dtype=dtype)
    stft = torch.stft(x, N_FFT, HOP, window=window, return_complex=True)
    mag = stft[..., :-1].abs() ** 2
    filt = torch.from_numpy(mel_filterbank(n_mels)).to(dtype)
    mel = filt @ mag
    log_spec = torch.clamp(mel, min=1e-10).log10()
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    log_spec = (log_spec + 4.0) / 4.0
    return log_spec[:, :N_FRAMES].to(torch.float32).numpy()[None]


# --------------------------------------------------------------------------------------------
# model math: the ONNX graphs run at core.py:826 / core.py:949 / align.py:616 are exports of
# HF WhisperModel (modeling_whisper.py); weights arrive as that state dict.
# --------------------------------------------------------------------------------------------
class OracleModel:
    def __init__(self, sd: Dict[str, torch.Tensor], n_heads: int, dtype=torch.float32):
        self.sd = {k: v.to(dtype) for k, v in sd.items()}
        self.H = n_heads
        self.d = self.sd["model.decoder.embed_positions.weight"].shape[1]
        self.Le = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("model.encoder.layers."))
        self.Ld = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("model.decoder.layers."))
        self.dtype = dtype

    # ---- shared blocks ----
    def _ln(self, x, p):
        return F.layer_norm(x, (self.d,), self.sd[p + ".weight"], self.sd[p + ".bias"], 1e-5)

    def _lin(self, x, p, bias=True):
        return F.linear(x, self.sd[p + ".weight"], self.sd[p + ".bias"] if bias else None)

    def _heads(self, x):  # [B,T,d] -> [B,H,T,64]
        B, T, _ = x.shape
        return x.view(B, T, self.H, self.d // self.H).transpose(1, 2)

    def _attn(self, xq, xkv, p, mask=None):
        """q scaled by head_dim^-0.5 BEFORE the product, k without bias (modeling_whisper.py:309-354)."""
        hd = self.d // self.H
        q = self._heads(self._lin(xq, p + ".q_proj") * hd ** -0.5)
        k = self._heads(self._lin(xkv, p + ".k_proj", bias=False))
        v = self._heads(self._lin(xkv, p + ".v_proj"))
        w = q @ k.transpose(2, 3)
        if mask is not None:
            w = w + mask
        w = torch.softmax(w, dim=-1)
        o = (w @ v).transpose(1, 2).reshape(xq.shape)
        return self._lin(o, p + ".out_proj"), w

    def _mlp(self, x, p):
        return self._lin(F.gelu(self._lin(x, p + ".fc1")), p + ".fc2")

    # ---- encoder: core.py:824-830 ----
    def encode(self, mel: np.ndarray) -> np.ndarray:
        """f32[B,n_mels,3000] -> f32[B,1500,d]."""
        with torch.no_grad():
            sd = self.sd
            x = torch.from_numpy(mel).to(self.dtype)
            x = F.gelu(F.conv1d(x, sd["model.encoder.conv1.weight"], sd["model.encoder.conv1.bias"], padding=1))
            x = F.gelu(F.conv1d(x, sd["model.encoder.conv2.weight"], sd["model.encoder.conv2.bias"], stride=2, padding=1))
            x = x.permute(0, 2, 1) + sd["model.encoder.embed_positions.weight"]
            for i in range(self.Le):
                p = f"model.encoder.layers.{i}"
                a, _ = self._attn(self._ln(x, p + ".self_attn_layer_norm"), self._ln(x, p + ".self_attn_layer_norm"),
                                  p + ".self_attn")
                x = x + a
                x = x + self._mlp(self._ln(x, p + ".final_layer_norm"), p)
            x = self._ln(x, "model.encoder.layer_norm")
            return x.to(torch.float32).numpy()

    # ---- decoder, cache-less full pass: core.py:941-949, align.py:608-616 ----
    def decode_full(self, ids: Sequence[int], enc: np.ndarray, want_attn: bool = False):
        """input_ids [1,T] + encoder_hidden_states [1,1500,d] -> logits f32[1,T,V]
        (+ per-layer cross-attention probabilities f32[1,H,T,1500])."""
        with torch.no_grad():
            sd = self.sd
            ids_t = torch.as_tensor(list(ids), dtype=torch.long)[None]
            T = ids_t.shape[1]
            enc_t = torch.from_numpy(np.asarray(enc, dtype=np.float32)).to(self.dtype)
            x = sd["model.decoder.embed_tokens.weight"][ids_t] + sd["model.decoder.embed_positions.weight"][:T]
            mask = torch.full((T, T), float("-inf"), dtype=self.dtype).triu(1)
            cross = []
            for i in range(self.Ld):
                p = f"model.decoder.layers.{i}"
                h = self._ln(x, p + ".self_attn_layer_norm")
                a, _ = self._attn(h, h, p + ".self_attn", mask)
                x = x + a
                a, w = self._attn(self._ln(x, p + ".encoder_attn_layer_norm"), enc_t, p + ".encoder_attn")
                x = x + a
                cross.append(w.to(torch.float32).numpy())
                x = x + self._mlp(self._ln(x, p + ".final_layer_norm"), p)
            x = self._ln(x, "model.decoder.layer_norm")
            # the output projection is tied to the token embedding unless the state dict carries its own copy
            proj = sd.get("proj_out.weight", sd["model.decoder.embed_tokens.weight"])
            logits = F.linear(x, proj).to(torch.float32).numpy()
        return (logits, cross) if want_attn else logits


class CachedDecoder:
    """The same decoder math with the schedule an optimised CPU implementation would use: cross-attention K/V projected
    once per window, self-attention K/V cached, one token per step.  NOT the reference's schedule (core.py:938-968 re-runs
    the whole prefix every step); it exists so that the CPU baseline can also be quoted for the better schedule, and as
    a cross-check of `decode_full` (same logits, same last-row cross-attention)."""

    def __init__(self, model: OracleModel, enc: np.ndarray):
        self.m = model
        m, sd = model, model.sd
        with torch.no_grad():
            enc_t = torch.from_numpy(np.asarray(enc, dtype=np.float32)).to(m.dtype)
            self.cross = []
            for i in range(m.Ld):
                p = f"model.decoder.layers.{i}.encoder_attn"
                self.cross.append((m._heads(m._lin(enc_t, p + ".k_proj", bias=False)), m._heads(m._lin(enc_t, p + ".v_proj"))))
        self.k = [None] * m.Ld
        self.v = [None] * m.Ld
        self.n = 0

    def forward(self, ids: Sequence[int]):
        """Appends `ids` at positions n .. n+len-1; returns (logits f32[len, V], per-layer cross-attention f32[H, len, 1500])."""
        m, sd = self.m, self.m.sd
        hd = m.d // m.H
        with torch.no_grad():
            ids_t = torch.as_tensor(list(ids), dtype=torch.long)[None]
            T = ids_t.shape[1]
            x = sd["model.decoder.embed_tokens.weight"][ids_t] + sd["model.decoder.embed_positions.weight"][self.n:self.n + T]
            total = self.n + T
            mask = torch.full((T, total), float("-inf"), dtype=m.dtype).triu(self.n + 1)
            cross_w = []
            for i in range(m.Ld):
                p = f"model.decoder.layers.{i}"
                h = m._ln(x, p + ".self_attn_layer_norm")
                q = m._heads(m._lin(h, p + ".self_attn.q_proj") * hd ** -0.5)
                k = m._heads(m._lin(h, p + ".self_attn.k_proj", bias=False))
                v = m._heads(m._lin(h, p + ".self_attn.v_proj"))
                self.k[i] = k if self.k[i] is None else torch.cat([self.k[i], k], dim=2)
                self.v[i] = v if self.v[i] is None else torch.cat([self.v[i], v], dim=2)
                w = torch.softmax(q @ self.k[i].transpose(2, 3) + mask, dim=-1)
                x = x + m._lin((w @ self.v[i]).transpose(1, 2).reshape(x.shape), p + ".self_attn.out_proj")
                h = m._ln(x, p + ".encoder_attn_layer_norm")
                q = m._heads(m._lin(h, p + ".encoder_attn.q_proj") * hd ** -0.5)
                ck, cv = self.cross[i]
                w = torch.softmax(q @ ck.transpose(2, 3), dim=-1)
                cross_w.append(w[0].to(torch.float32).numpy())
                x = x + m._lin((w @ cv).transpose(1, 2).reshape(x.shape), p + ".encoder_attn.out_proj")
                x = x + m._mlp(m._ln(x, p + ".final_layer_norm"), p)
            x = m._ln(x, "model.decoder.layer_norm")
            proj = sd.get("proj_out.weight", sd["model.decoder.embed_tokens.weight"])
            logits = F.linear(x, proj)[0].to(torch.float32).numpy()
        self.n = total
        return logits, cross_w


def greedy_reference_schedule(model: OracleModel, enc: np.ndarray, prompt_ids: Sequence[int],
                              max_tokens: int, eot: int) -> List[int]:
    """core.py:938-968: every step re-runs the whole decoder over the whole prefix, takes
    np.argmax of the last position (first index on ties), stops on eot without appending."""
    cur = list(prompt_ids)
    new: List[int] = []
    for _ in range(max(0, max_tokens - len(prompt_ids))):
        logits = model.decode_full(cur, enc)
        nxt = int(np.argmax(logits[0, -1, :]))
        if nxt == eot:
            break
        cur.append(nxt)
        new.append(nxt)
    return new


def filter_text_tokens(tokens: Sequence[int], sot: int, ts_begin: int, ts_end: int) -> List[int]:
    """core.py:991."""
    return [t for t in tokens if t < sot and not (ts_begin <= t <= ts_end)]


def attention_rows_reference_schedule(model: OracleModel, enc: np.ndarray, prompt_ids: Sequence[int],
                                      text_ids: Sequence[int]) -> List[np.ndarray]:
    """align.py:603-630: one full decoder pass per text token over prompt + text[:i], keeping the
    LAST position's cross-attention per layer -> per layer f32[1,H,N,1500]."""
    per_layer: List[List[np.ndarray]] = [[] for _ in range(model.Ld)]
    cur = list(prompt_ids)
    for tok in text_ids:
        _, cross = model.decode_full(cur, enc, want_attn=True)
        for l, w in enumerate(cross):
            per_layer[l].append(w[0, :, -1, :])
        cur.append(tok)
    return [np.stack(rows, axis=1)[None] for rows in per_layer]


def attention_rows_teacher_forced(model: OracleModel, enc: np.ndarray, prompt_ids: Sequence[int],
                                  text_ids: Sequence[int]) -> List[np.ndarray]:
    """Same rows as `attention_rows_reference_schedule` from ONE pass over prompt + text[:-1]
    (causality makes the two identical); used where the N-pass form is too slow."""
    seq = list(prompt_ids) + list(text_ids[:-1])
    P, N = len(prompt_ids), len(text_ids)
    _, cross = model.decode_full(seq, enc, want_attn=True)
    return [w[:, :, P - 1:P - 1 + N, :] for w in cross]


# --------------------------------------------------------------------------------------------
# alignment numerics: align.py:717-853
# --------------------------------------------------------------------------------------------
def cost_matrix(cross_attentions: List[np.ndarray], heads: Sequence[Tuple[int, int]],
                n_frames: int = N_AUDIO_CTX, medfilt_width: int = 7) -> np.ndarray:
    """align.py:725-811,818: stack layers, take the selected (layer, head) rows, mean over heads (f32),
    scipy median_filter (1,7) (default mode 'reflect'), softmax over frames, negate, cast to f64."""
    from scipy.ndimage import median_filter
    w = torch.stack([torch.from_numpy(a) for a in cross_attentions]).squeeze(1)   # [L,H,N,S]
    sel = [w[l, h, :, :n_frames] for l, h in heads]
    m = torch.stack(sel).mean(dim=0).float().numpy()
    if medfilt_width > 0 and m.shape[1] > medfilt_width:
        m = median_filter(m, (1, medfilt_width))
    m = torch.from_numpy(m * 1.0).softmax(dim=-1).numpy()
    return (-m).astype(np.double)


def dtw_symmetric1(cost: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """dtw.dtw(cost, step_pattern=symmetric1, keep_internals=False) (align.py:818-831).

    dtw-python 1.5.3, restated from its published algorithm: the pattern lists, in order,
    (i-1,j-1), (i,j-1), (i-1,j), each with local-cost weight 1; the cumulative matrix is
    D[0,0] = c[0,0], D[i,j] = min_p (D[prev_p] + c[i,j]) where each candidate SUM is formed first and
    the minimum is taken with a strict '<' scan in pattern order (ties keep the earliest pattern,
    unreachable predecessors are skipped); the direction matrix records the winning pattern and the
    path is backtracked from (N-1, M-1) to (0,0).  Returns (index1, index2) ascending."""
    c = np.ascontiguousarray(cost, dtype=np.float64)
    n, m = c.shape
    D = np.full((n, m), np.nan)
    S = np.zeros((n, m), dtype=np.int8)
    D[0, 0] = c[0, 0]
    for j in range(m):
        for i in range(n):
            if i == 0 and j == 0:
                continue
            best, arg = np.inf, 0
            cij = c[i, j]
            if i > 0 and j > 0:
                v = D[i - 1, j - 1] + cij
                if v < best:
                    best, arg = v, 1
            if j > 0:
                v = D[i, j - 1] + cij
                if v < best:
                    best, arg = v, 2
            if i > 0:
                v = D[i - 1, j] + cij
                if v < best:
                    best, arg = v, 3
            if arg:
                D[i, j] = best
                S[i, j] = arg
    if np.isnan(D[n - 1, m - 1]):
        raise ValueError("No warping path found compatible with the local constraints")
    i, j = n - 1, m - 1
    p1, p2 = [i], [j]
    while not (i == 0 and j == 0):
        s = S[i, j]
        if s == 1:
            i, j = i - 1, j - 1
        elif s == 2:
            j -= 1
        else:
            i -= 1
        p1.append(i)
        p2.append(j)
    return np.array(p1[::-1], dtype=np.int64), np.array(p2[::-1], dtype=np.int64)


def dtw_bruteforce_cost(cost: np.ndarray) -> float:
    """Minimum total cost over every monotone path with steps (1,1),(0,1),(1,0): recursion written
    independently of `dtw_symmetric1` (memoised DFS), for small matrices only."""
    from functools import lru_cache
    c = np.asarray(cost, dtype=np.float64)
    n, m = c.shape

    @lru_cache(maxsize=None)
    def best(i, j):
        if i == 0 and j == 0:
            return float(c[0, 0])
        cands = []
        if i > 0 and j > 0:
            cands.append(best(i - 1, j - 1))
        if j > 0:
            cands.append(best(i, j - 1))
        if i > 0:
            cands.append(best(i - 1, j))
        return min(cands) + float(c[i, j])

    return best(n - 1, m - 1)


def token_boundaries(index1: np.ndarray, index2: np.ndarray, n_tokens: int,
                     n_frames: int = N_AUDIO_CTX) -> np.ndarray:
    """align.py:835-853: frame at every jump of the token index, a 0 in front, then the length is
    forced to n_tokens + 1 (padding with the last value / truncating) and the last entry clamped to
    n_frames.  The un-fixed array has n_tokens entries, so the pad branch always runs."""
    change = np.where(np.diff(index1) > 0)[0] + 1
    b = np.pad(index2[change], (1, 0), constant_values=0)
    expected = n_tokens + 1
    if len(b) != expected:
        if len(b) < expected:
            last = b[-1] if len(b) > 0 else n_frames
            b = np.pad(b, (0, expected - len(b)), constant_values=last)
        else:
            b = b[:expected]
        b[-1] = min(b[-1], n_frames)
    return b.astype(np.int64)


# --------------------------------------------------------------------------------------------
# whole-window reference schedule (what the CPU baseline times): core.py:814-1030 numerics only
# --------------------------------------------------------------------------------------------
def transcribe_window_reference(model: OracleModel, audio: np.ndarray, prompt_ids: Sequence[int],
                                heads: Sequence[Tuple[int, int]], max_tokens: int, eot: int, sot: int,
                                ts_begin: int, ts_end: int, n_mels: int = 80,
                                teacher_forced_rows: bool = False):
    """mel -> encoder -> cache-less greedy loop -> text filter -> per-token attention replay ->
    cost matrix -> DTW -> token boundary frames.  Returns (new_tokens, text_ids, boundaries)."""
    mel = log_mel_segment(audio, n_mels)
    enc = model.encode(mel)
    new = greedy_reference_schedule(model, enc, prompt_ids, max_tokens, eot)
    text = filter_text_tokens(new, sot, ts_begin, ts_end)
    if not text:
        return new, text, np.zeros((0,), dtype=np.int64)
    rows = (attention_rows_teacher_forced if teacher_forced_rows else attention_rows_reference_schedule)(
        model, enc, prompt_ids, text)
    cost = cost_matrix(rows, heads)
    i1, i2 = dtw_symmetric1(cost)
    return new, text, token_boundaries(i1, i2, len(text))
