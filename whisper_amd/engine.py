"""Host-side owner of one libuw_hip context: uploads weights, holds device tensors, exposes the stages of
the transcription path (mel, encoder, cross K/V, decoder, greedy decode with attention-row capture, cost
matrix, DTW) as methods over torch CUDA tensors.  PyTorch is used for device memory and streams only."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as L
from .config import N_AUDIO_CTX, N_FRAMES, WhisperDims, alignment_heads
from .melbank import mel_filterbank
from .weights import check_state_dict, tensor_plan


def _i32(x, device) -> torch.Tensor:
    return torch.as_tensor(np.asarray(x, dtype=np.int32), device=device)


class Engine:
    def __init__(self, dims: WhisperDims, state_dict: Dict[str, torch.Tensor], dtype: str = "bf16", max_batch: int = 1,
                 align_heads: Optional[Sequence[Tuple[int, int]]] = None, device: str = "cuda:0",
                 n_text_ctx: Optional[int] = None):
        if not torch.cuda.is_available():
            raise L.UwError("no GPU visible: the transcription path runs on MI355X only (no CPU fallback)")
        self.lib = L.load()
        self.dims = dims
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.dtype = dtype
        self.max_batch = int(max_batch)
        self.heads = list(align_heads) if align_heads is not None else []
        self.n_text_ctx = int(n_text_ctx or dims.n_text_ctx)
        check_state_dict(dims, state_dict)
        cfg = L.UwConfig()
        cfg.d_model, cfg.n_heads, cfg.enc_layers, cfg.dec_layers = dims.d_model, dims.n_heads, dims.enc_layers, dims.dec_layers
        cfg.vocab, cfg.n_mels, cfg.n_audio_ctx, cfg.n_text_ctx = dims.vocab, dims.n_mels, dims.n_audio_ctx, self.n_text_ctx
        cfg.dtype = {"f32": L.UW_DTYPE_F32, "fp32": L.UW_DTYPE_F32, "bf16": L.UW_DTYPE_BF16}[dtype]
        cfg.max_batch = self.max_batch
        if len(self.heads) > L.UW_MAX_ALIGN_HEADS:
            raise ValueError("too many alignment heads")
        cfg.n_align_heads = len(self.heads)
        for k, (l, h) in enumerate(self.heads):
            cfg.align_layer[k], cfg.align_head[k] = int(l), int(h)
        plan = tensor_plan(dims)
        dev_w = []
        for name, _, _ in plan:
            t = state_dict[name]
            if self.n_text_ctx != dims.n_text_ctx and name.endswith("decoder.embed_positions.weight"):
                t = t[: self.n_text_ctx]
            dev_w.append(t.to(self.device, torch.float32).contiguous())
        arr = (C.c_void_p * len(dev_w))(*[t.data_ptr() for t in dev_w])
        filt = np.ascontiguousarray(mel_filterbank(dims.n_mels), dtype=np.float32)
        handle = C.c_void_p()
        torch.cuda.synchronize()
        L.check(self.lib.uw_ctx_create(C.byref(cfg), arr, len(dev_w), filt.ctypes.data_as(C.c_void_p), C.byref(handle)))
        self.ctx = handle
        del dev_w
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "ctx", None):
            self.lib.uw_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_bytes(self) -> int:
        return int(self.lib.uw_ctx_bytes(self.ctx))

    def _s(self) -> int:
        return L.stream_handle()

    def profile_begin(self, max_launches: int = 16384) -> None:
        L.check(self.lib.uw_profile_begin(self.ctx, int(max_launches)))

    def profile_end(self) -> Dict[str, float]:
        total, timed, windows, ms = C.c_int64(), C.c_int64(), C.c_int64(), C.c_double()
        L.check(self.lib.uw_profile_end(self.ctx, C.byref(total), C.byref(timed), C.byref(windows), C.byref(ms)))
        out = {"launches": total.value, "timed": timed.value, "windows": windows.value, "timed_ms": ms.value}
        for kind, name in ((1, "self_decode"), (2, "encoder"), (3, "dtw"), (4, "decode_step")):
            L.check(self.lib.uw_profile_span(self.ctx, kind, C.byref(total), C.byref(timed), C.byref(ms)))
            out[name] = {"launches": total.value, "timed": timed.value, "timed_ms": ms.value}
        return out

    # ---- stages ---------------------------------------------------------------------------------
    def mel(self, audio: torch.Tensor, n_samples: torch.Tensor) -> torch.Tensor:
        """audio f32 [B, stride] (cuda), n_samples i32 [B] -> log-mel f32 [B, n_mels, 3000]."""
        assert audio.is_cuda and audio.dtype == torch.float32 and audio.dim() == 2 and audio.is_contiguous()
        B, stride = audio.shape
        n_samples = n_samples.to(self.device, torch.int32).contiguous()
        out = torch.empty((B, self.dims.n_mels, N_FRAMES), dtype=torch.float32, device=self.device)
        L.check(self.lib.uw_mel(self.ctx, L.ptr(audio), stride, L.ptr(n_samples), B, L.ptr(out), self._s()))
        return out

    def encode(self, mel: torch.Tensor) -> torch.Tensor:
        assert mel.is_cuda and mel.dtype == torch.float32 and mel.is_contiguous()
        B = mel.shape[0]
        assert tuple(mel.shape[1:]) == (self.dims.n_mels, N_FRAMES), mel.shape
        out = torch.empty((B, N_AUDIO_CTX, self.dims.d_model), dtype=torch.float32, device=self.device)
        L.check(self.lib.uw_encode(self.ctx, L.ptr(mel), B, L.ptr(out), self._s()))
        return out

    def cross_kv(self, B: int, enc_states: Optional[torch.Tensor] = None) -> None:
        if enc_states is not None:
            assert enc_states.is_cuda and enc_states.dtype == torch.float32 and enc_states.is_contiguous()
            assert tuple(enc_states.shape) == (B, N_AUDIO_CTX, self.dims.d_model)
        L.check(self.lib.uw_cross_kv(self.ctx, L.ptr(enc_states), B, self._s()))

    def decoder_forward(self, ids: torch.Tensor, logits_all: bool = False, logits_last: bool = True,
                        cross_probs: bool = False):
        """ids i32 [B, T] -> dict with the requested outputs (cache-less decoder pass of the reference)."""
        ids = ids.to(self.device, torch.int32).contiguous()
        B, T = ids.shape
        out = {}
        la = torch.empty((B, T, self.dims.vocab), dtype=torch.float32, device=self.device) if logits_all else None
        ll = torch.empty((B, self.dims.vocab), dtype=torch.float32, device=self.device) if logits_last else None
        cp = (torch.empty((self.dims.dec_layers, B, self.dims.n_heads, T, N_AUDIO_CTX), dtype=torch.float32,
                          device=self.device) if cross_probs else None)
        L.check(self.lib.uw_decoder_forward(self.ctx, L.ptr(ids), T, B, T, L.ptr(la), L.ptr(ll), L.ptr(cp), self._s()))
        if la is not None:
            out["logits_all"] = la
        if ll is not None:
            out["logits_last"] = ll
        if cp is not None:
            out["cross_probs"] = cp
        return out

    def decode_step(self, ids: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
        """KV-cached single step: token ids[b] at position pos[b] -> logits f32 [B, vocab].  Earlier positions must
        already be in the self-attention cache (decoder_forward over the prefix, or previous steps)."""
        ids = ids.to(self.device, torch.int32).contiguous()
        pos = pos.to(self.device, torch.int32).contiguous()
        B = ids.shape[0]
        out = torch.empty((B, self.dims.vocab), dtype=torch.float32, device=self.device)
        L.check(self.lib.uw_decode_step(self.ctx, L.ptr(ids), L.ptr(pos), B, L.ptr(out), self._s()))
        return out

    def _pack_prompts(self, prompts: Sequence[Sequence[int]]):
        B = len(prompts)
        lens = np.array([len(p) for p in prompts], dtype=np.int32)
        ld = int(max(1, lens.max()))
        host = np.zeros((B, ld), dtype=np.int32)
        for b, p in enumerate(prompts):
            host[b, : len(p)] = np.asarray(p, dtype=np.int32)
        return torch.from_numpy(host).to(self.device), lens, ld

    def decode_greedy(self, prompts: Sequence[Sequence[int]], max_tokens: int, eot: int, sot: int, ts_begin: int,
                      ts_end: int) -> Dict[str, torch.Tensor]:
        B = len(prompts)
        ids, lens, ld = self._pack_prompts(prompts)
        T = self.n_text_ctx
        new_tokens = torch.zeros((B, T), dtype=torch.int32, device=self.device)
        text_ids = torch.zeros((B, T), dtype=torch.int32, device=self.device)
        n_new = torch.zeros((B,), dtype=torch.int32, device=self.device)
        n_text = torch.zeros((B,), dtype=torch.int32, device=self.device)
        L.check(self.lib.uw_decode_greedy(self.ctx, L.ptr(ids), ld, lens.ctypes.data_as(C.POINTER(C.c_int32)), B,
                                          int(max_tokens), int(eot), int(sot), int(ts_begin), int(ts_end),
                                          L.ptr(new_tokens), L.ptr(n_new), L.ptr(text_ids), L.ptr(n_text), self._s()))
        return {"new_tokens": new_tokens, "n_new": n_new, "text_ids": text_ids, "n_text": n_text, "status": self.window_status(B)}

    def window_status(self, B: int) -> torch.Tensor:
        """Per window of the last greedy decode: 0 = decoded normally, 1 = stopped at a non-finite logits row with the
        tokens before it kept (the reference's decoder-run failure inside its generation loop, core.py:948-980)."""
        out = np.zeros((B,), dtype=np.int32)
        L.check(self.lib.uw_window_status(self.ctx, B, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return torch.from_numpy(out)          # host tensor: the library hands the flags over on the host

    def attn_rows_forced(self, prompts: Sequence[Sequence[int]], texts: Sequence[Sequence[int]]) -> None:
        B = len(prompts)
        pids, plens, pld = self._pack_prompts(prompts)
        tids, tlens, tld = self._pack_prompts(texts)
        L.check(self.lib.uw_attn_rows_forced(self.ctx, L.ptr(pids), pld, plens.ctypes.data_as(C.POINTER(C.c_int32)),
                                             L.ptr(tids), tld, tlens.ctypes.data_as(C.POINTER(C.c_int32)), B, self._s()))

    def attn_rows(self, B: int, n_rows: int) -> torch.Tensor:
        out = torch.empty((B, len(self.heads), n_rows, N_AUDIO_CTX), dtype=torch.float32, device=self.device)
        L.check(self.lib.uw_get_attn_rows(self.ctx, B, n_rows, L.ptr(out), self._s()))
        return out

    def align(self, n_text: torch.Tensor) -> torch.Tensor:
        n_text = n_text.to(self.device, torch.int32).contiguous()
        B = n_text.shape[0]
        out = torch.zeros((B, self.n_text_ctx + 1), dtype=torch.int32, device=self.device)
        L.check(self.lib.uw_align(self.ctx, L.ptr(n_text), B, L.ptr(out), self._s()))
        return out

    def transcribe_windows(self, audio: torch.Tensor, n_samples: torch.Tensor, prompts: Sequence[Sequence[int]],
                           max_tokens: int, eot: int, sot: int, ts_begin: int, ts_end: int) -> Dict[str, torch.Tensor]:
        """The whole path for a batch of windows; returns device tensors (tokens, counts, boundary frames)."""
        assert audio.is_cuda and audio.dtype == torch.float32 and audio.dim() == 2 and audio.is_contiguous()
        B, stride = audio.shape
        assert len(prompts) == B
        n_samples = n_samples.to(self.device, torch.int32).contiguous()
        ids, lens, ld = self._pack_prompts(prompts)
        T = self.n_text_ctx
        new_tokens = torch.zeros((B, T), dtype=torch.int32, device=self.device)
        text_ids = torch.zeros((B, T), dtype=torch.int32, device=self.device)
        n_new = torch.zeros((B,), dtype=torch.int32, device=self.device)
        n_text = torch.zeros((B,), dtype=torch.int32, device=self.device)
        bounds = torch.zeros((B, T + 1), dtype=torch.int32, device=self.device)
        L.check(self.lib.uw_transcribe_windows(
            self.ctx, L.ptr(audio), stride, L.ptr(n_samples), L.ptr(ids), ld, lens.ctypes.data_as(C.POINTER(C.c_int32)),
            B, int(max_tokens), int(eot), int(sot), int(ts_begin), int(ts_end), L.ptr(new_tokens), L.ptr(n_new),
            L.ptr(text_ids), L.ptr(n_text), L.ptr(bounds), self._s()))
        return {"new_tokens": new_tokens, "n_new": n_new, "text_ids": text_ids, "n_text": n_text, "boundaries": bounds,
                "status": self.window_status(B)}


# ---- context-free alignment kernels ----------------------------------------------------------------
def cost_matrix(rows: torch.Tensor, n_tok: torch.Tensor) -> torch.Tensor:
    """rows f32 [B, n_sel, N, S] (cuda) -> f64 cost [B, N, S] (align.py:758-811,818)."""
    lib = L.load()
    assert rows.is_cuda and rows.dtype == torch.float32 and rows.is_contiguous() and rows.dim() == 4
    B, n_sel, N, S = rows.shape
    n_tok = n_tok.to(rows.device, torch.int32).contiguous()
    cost = torch.zeros((B, N, S), dtype=torch.float64, device=rows.device)
    L.check(lib.uw_cost_matrix(L.ptr(rows), n_sel, L.ptr(n_tok), B, N, S, L.ptr(cost), L.stream_handle()))
    return cost


def dtw(cost: torch.Tensor, n_tok: torch.Tensor, want_path: bool = True):
    """cost f64 [B, N, S] (cuda) -> (boundaries i32 [B, N+1], path_i, path_j, path_len) (align.py:818-853)."""
    lib = L.load()
    assert cost.is_cuda and cost.dtype == torch.float64 and cost.is_contiguous() and cost.dim() == 3
    B, N, S = cost.shape
    n_tok = n_tok.to(cost.device, torch.int32).contiguous()
    ws = torch.empty((int(lib.uw_dtw_workspace_bytes(B, N, S)),), dtype=torch.uint8, device=cost.device)
    bnd = torch.zeros((B, N + 1), dtype=torch.int32, device=cost.device)
    pi = torch.zeros((B, N + S), dtype=torch.int32, device=cost.device) if want_path else None
    pj = torch.zeros((B, N + S), dtype=torch.int32, device=cost.device) if want_path else None
    plen = torch.zeros((B,), dtype=torch.int32, device=cost.device)
    L.check(lib.uw_dtw(L.ptr(cost), L.ptr(n_tok), B, N, S, L.ptr(ws), L.ptr(bnd), L.ptr(pi), L.ptr(pj), L.ptr(plen),
                       L.stream_handle()))
    return bnd, pi, pj, plen
