"""Engine-shaped test double that computes with the CPU oracle (tests only: lets the host logic of
whisper_amd.core run without a GPU so that it can be compared with outputs of the reference's own core.py)."""
import numpy as np
import torch

from oracle import whisper_oracle as O
from whisper_amd.config import N_AUDIO_CTX, N_TEXT_CTX


class OracleEngine:
    """Faults (for the failure-isolation cases) are expressed the way the GPU engine shows them: a call that covers a
    window of `fail_windows` raises; a window of `no_align` comes back with boundaries[0] = -1 (no warping path); a window
    in `decode_stop` (window -> tokens kept) comes back truncated with status 1 (non-finite logits).  Windows are
    numbered in the order their audio is first seen, so a group that is retried window by window keeps its numbers."""

    def __init__(self, dims, sd, heads, max_batch=4, fail_windows=(), no_align=(), decode_stop=None):
        self.dims, self.heads, self.max_batch = dims, list(heads), max_batch
        self.device = torch.device("cpu")
        self.n_text_ctx = N_TEXT_CTX
        self.om = O.OracleModel(sd, dims.n_heads)
        self._enc = None
        self._rows = None
        self.fail_windows, self.no_align, self.decode_stop = set(fail_windows), set(no_align), dict(decode_stop or {})
        self._seen = {}

    def _window_ids(self, audio, n_samples):
        ids = []
        for b in range(audio.shape[0]):
            key = hash(audio[b, : int(n_samples[b])].numpy().tobytes())
            ids.append(self._seen.setdefault(key, len(self._seen) + 1))
        return ids

    # --- batched fast path -----------------------------------------------------------------------
    def transcribe_windows(self, audio, n_samples, prompts, max_tokens, eot, sot, ts_begin, ts_end):
        B = len(prompts)
        T = self.n_text_ctx
        wids = self._window_ids(audio, n_samples)
        if any(w in self.fail_windows for w in wids):
            raise RuntimeError(f"injected engine failure: the call covers windows {wids}")
        out = {"new_tokens": torch.zeros((B, T), dtype=torch.int32), "text_ids": torch.zeros((B, T), dtype=torch.int32),
               "n_new": torch.zeros(B, dtype=torch.int32), "n_text": torch.zeros(B, dtype=torch.int32),
               "boundaries": torch.zeros((B, T + 1), dtype=torch.int32), "status": torch.zeros(B, dtype=torch.int32)}
        for b in range(B):
            a = audio[b, : int(n_samples[b])].numpy()
            mel = O.log_mel_segment(a, self.dims.n_mels)
            enc = self.om.encode(mel)
            new = O.greedy_reference_schedule(self.om, enc, list(prompts[b]), max_tokens, eot)
            if wids[b] in self.decode_stop:
                new = new[: self.decode_stop[wids[b]]]
                out["status"][b] = 1
            text = O.filter_text_tokens(new, sot, ts_begin, ts_end)
            out["new_tokens"][b, : len(new)] = torch.tensor(new, dtype=torch.int32)
            out["text_ids"][b, : len(text)] = torch.tensor(text, dtype=torch.int32)
            out["n_new"][b], out["n_text"][b] = len(new), len(text)
            if wids[b] in self.no_align:
                out["boundaries"][b, 0] = -1
            elif text and self.heads:
                rows = O.attention_rows_reference_schedule(self.om, enc, list(prompts[b]), text)
                i1, i2 = O.dtw_symmetric1(O.cost_matrix(rows, self.heads))
                out["boundaries"][b, : len(text) + 1] = torch.from_numpy(O.token_boundaries(i1, i2, len(text)).astype(np.int32))
        return out

    # --- staged calls used by the forced-alignment mode -------------------------------------------
    def mel(self, audio, n_samples):
        self._stage_ids = self._window_ids(audio, n_samples)
        return torch.from_numpy(np.concatenate([O.log_mel_segment(audio[b, : int(n_samples[b])].numpy(), self.dims.n_mels)
                                                for b in range(audio.shape[0])], 0))

    def encode(self, mel):
        if any(w in self.fail_windows for w in self._stage_ids):
            raise RuntimeError(f"injected encoder failure: the call covers windows {self._stage_ids}")
        self._enc = [self.om.encode(mel[b: b + 1].numpy()) for b in range(mel.shape[0])]
        return torch.from_numpy(np.concatenate(self._enc, 0))

    def cross_kv(self, B, enc_states=None):
        pass

    def attn_rows_forced(self, prompts, texts):
        self._rows = [O.attention_rows_reference_schedule(self.om, self._enc[b], list(prompts[b]), list(texts[b]))
                      for b in range(len(prompts))]

    def align(self, n_text):
        B = len(self._rows)
        out = torch.zeros((B, self.n_text_ctx + 1), dtype=torch.int32)
        for b in range(B):
            n = int(n_text[b])
            i1, i2 = O.dtw_symmetric1(O.cost_matrix(self._rows[b], self.heads))
            out[b, : n + 1] = torch.from_numpy(O.token_boundaries(i1, i2, n).astype(np.int32))
        return out

    def close(self):
        pass


class FaultInjector:
    """Wraps a real engine and shows the same injected faults as OracleEngine does (see its docstring), so the
    failure-isolation goldens can also run on the GPU engine: `fail_windows` calls raise, `no_align` windows come back
    with boundaries[0] = -1, `decode_stop` windows are cut to the stated number of tokens with status 1."""

    def __init__(self, engine, fail_windows=(), no_align=(), decode_stop=None):
        self._eng = engine
        self.fail_windows, self.no_align, self.decode_stop = set(fail_windows), set(no_align), dict(decode_stop or {})
        self._seen = {}
        self._stage_ids = []

    def __getattr__(self, name):
        return getattr(self._eng, name)

    def _window_ids(self, audio, n_samples):
        a = audio.cpu()
        ids = []
        for b in range(a.shape[0]):
            key = hash(a[b, : int(n_samples[b])].numpy().tobytes())
            ids.append(self._seen.setdefault(key, len(self._seen) + 1))
        return ids

    def transcribe_windows(self, audio, n_samples, prompts, max_tokens, eot, sot, ts_begin, ts_end):
        wids = self._window_ids(audio, n_samples)
        if any(w in self.fail_windows for w in wids):
            raise RuntimeError(f"injected engine failure: the call covers windows {wids}")
        res = {k: v.clone() for k, v in self._eng.transcribe_windows(audio, n_samples, prompts, max_tokens, eot, sot, ts_begin, ts_end).items()}
        for b, w in enumerate(wids):
            if w in self.decode_stop:
                keep = self.decode_stop[w]
                new = res["new_tokens"][b, :keep].tolist()
                text = O.filter_text_tokens(new, sot, ts_begin, ts_end)
                res["new_tokens"][b, keep:] = 0
                res["text_ids"][b] = 0
                res["text_ids"][b, : len(text)] = torch.tensor(text, dtype=torch.int32)
                res["n_new"][b], res["n_text"][b], res["status"][b] = keep, len(text), 1
            if w in self.no_align:
                res["boundaries"][b, 0] = -1
        return res

    def mel(self, audio, n_samples):
        self._stage_ids = self._window_ids(audio, n_samples)
        return self._eng.mel(audio, n_samples)

    def encode(self, mel):
        if any(w in self.fail_windows for w in self._stage_ids):
            raise RuntimeError(f"injected encoder failure: the call covers windows {self._stage_ids}")
        return self._eng.encode(mel)
