"""Engine-shaped test double that computes with the CPU oracle (tests only: lets the host logic of
whisper_amd.core run without a GPU so that it can be compared with outputs of the reference's own core.py)."""
import numpy as np
import torch

from oracle import whisper_oracle as O
from whisper_amd.config import N_AUDIO_CTX, N_TEXT_CTX


class OracleEngine:
    def __init__(self, dims, sd, heads, max_batch=4):
        self.dims, self.heads, self.max_batch = dims, list(heads), max_batch
        self.device = torch.device("cpu")
        self.n_text_ctx = N_TEXT_CTX
        self.om = O.OracleModel(sd, dims.n_heads)
        self._enc = None
        self._rows = None

    # --- batched fast path -----------------------------------------------------------------------
    def transcribe_windows(self, audio, n_samples, prompts, max_tokens, eot, sot, ts_begin, ts_end):
        B = len(prompts)
        T = self.n_text_ctx
        out = {"new_tokens": torch.zeros((B, T), dtype=torch.int32), "text_ids": torch.zeros((B, T), dtype=torch.int32),
               "n_new": torch.zeros(B, dtype=torch.int32), "n_text": torch.zeros(B, dtype=torch.int32),
               "boundaries": torch.zeros((B, T + 1), dtype=torch.int32)}
        for b in range(B):
            a = audio[b, : int(n_samples[b])].numpy()
            mel = O.log_mel_segment(a, self.dims.n_mels)
            enc = self.om.encode(mel)
            new = O.greedy_reference_schedule(self.om, enc, list(prompts[b]), max_tokens, eot)
            text = O.filter_text_tokens(new, sot, ts_begin, ts_end)
            out["new_tokens"][b, : len(new)] = torch.tensor(new, dtype=torch.int32)
            out["text_ids"][b, : len(text)] = torch.tensor(text, dtype=torch.int32)
            out["n_new"][b], out["n_text"][b] = len(new), len(text)
            if text and self.heads:
                rows = O.attention_rows_reference_schedule(self.om, enc, list(prompts[b]), text)
                i1, i2 = O.dtw_symmetric1(O.cost_matrix(rows, self.heads))
                out["boundaries"][b, : len(text) + 1] = torch.from_numpy(O.token_boundaries(i1, i2, len(text)).astype(np.int32))
        return out

    # --- staged calls used by the forced-alignment mode -------------------------------------------
    def mel(self, audio, n_samples):
        return torch.from_numpy(np.concatenate([O.log_mel_segment(audio[b, : int(n_samples[b])].numpy(), self.dims.n_mels)
                                                for b in range(audio.shape[0])], 0))

    def encode(self, mel):
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
