"""onnxruntime-shaped session objects over the HIP engine.

urro_whisper drives its numerics through two `InferenceSession` objects (urro_whisper/model/load.py:70,86) and
only ever touches `get_inputs()[i].name`, `get_outputs()[i].name`, `get_providers()` and
`run(output_names | None, feeds) -> list[np.ndarray]` (core.py:183-192, 824-830, 941-949; align.py:608-618).
These classes provide exactly that surface, so the reference's own chunk loop can run on MI355X unchanged;
the batched fast path (`Engine.transcribe_windows`) is what whisper_amd.core uses instead.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

PROVIDER = "MI355XExecutionProvider"


class _Arg:
    def __init__(self, name: str, shape, type_: str):
        self.name, self.shape, self.type = name, shape, type_


class EncoderSession:
    def __init__(self, engine):
        self.engine = engine

    def get_providers(self) -> List[str]:
        return [PROVIDER]

    def get_inputs(self):
        return [_Arg("input_features", ["batch", self.engine.dims.n_mels, 3000], "tensor(float)")]

    def get_outputs(self):
        return [_Arg("last_hidden_state", ["batch", 1500, self.engine.dims.d_model], "tensor(float)")]

    def run(self, output_names: Optional[Sequence[str]], feeds: Dict[str, np.ndarray]):
        name = self.get_inputs()[0].name
        if name not in feeds:
            raise ValueError(f"Required inputs (['{name}']) are missing from input feed ({list(feeds)}).")
        mel = np.ascontiguousarray(feeds[name], dtype=np.float32)
        if mel.ndim != 3 or mel.shape[1:] != (self.engine.dims.n_mels, 3000):
            raise ValueError(f"input_features must be [batch, {self.engine.dims.n_mels}, 3000], got {mel.shape}")
        if mel.shape[0] > self.engine.max_batch:
            raise ValueError(f"batch {mel.shape[0]} exceeds the engine's max_batch {self.engine.max_batch}")
        out = self.engine.encode(torch.from_numpy(mel).to(self.engine.device))
        return [out.cpu().numpy()]


class DecoderSession:
    """Cache-less decoder graph: logits and per-layer cross-attention probabilities for the whole prefix.
    Advertises no past_key_values inputs, so the reference skips its dummy-cache feeds (core.py:840-871)."""

    def __init__(self, engine):
        self.engine = engine
        self._enc_key = None

    def get_providers(self) -> List[str]:
        return [PROVIDER]

    def get_inputs(self):
        d = self.engine.dims.d_model
        return [_Arg("input_ids", ["batch", "sequence"], "tensor(int64)"),
                _Arg("encoder_hidden_states", ["batch", 1500, d], "tensor(float)")]

    def get_outputs(self):
        dims = self.engine.dims
        outs = [_Arg("logits", ["batch", "sequence", dims.vocab], "tensor(float)")]
        outs += [_Arg(f"cross_attentions.{l}", ["batch", dims.n_heads, "sequence", 1500], "tensor(float)")
                 for l in range(dims.dec_layers)]
        return outs

    def _ensure_cross_kv(self, enc: np.ndarray) -> None:
        # the reference passes the same encoder states on every step of a chunk: project them once
        key = (enc.shape, enc.__array_interface__["data"][0], float(enc.flat[0]), float(enc.flat[-1]), float(enc.sum()))
        if key != self._enc_key:
            self.engine.cross_kv(enc.shape[0], torch.from_numpy(enc).to(self.engine.device))
            self._enc_key = key

    def run(self, output_names: Optional[Sequence[str]], feeds: Dict[str, np.ndarray]):
        missing = [n for n in ("input_ids", "encoder_hidden_states") if n not in feeds]
        if missing:
            raise ValueError(f"Required inputs ({missing}) are missing from input feed ({list(feeds)}).")
        ids = np.asarray(feeds["input_ids"])
        enc = np.ascontiguousarray(feeds["encoder_hidden_states"], dtype=np.float32)
        if ids.ndim != 2 or enc.ndim != 3 or ids.shape[0] != enc.shape[0]:
            raise ValueError(f"bad decoder feed shapes {ids.shape} / {enc.shape}")
        names = [o.name for o in self.get_outputs()] if output_names is None else list(output_names)
        known = {o.name for o in self.get_outputs()}
        for n in names:
            if n not in known:
                raise ValueError(f"Invalid output name: {n}")
        self._ensure_cross_kv(enc)
        want_probs = any(n.startswith("cross_attentions") for n in names)
        out = self.engine.decoder_forward(torch.from_numpy(ids.astype(np.int32)), logits_all="logits" in names,
                                          logits_last=False, cross_probs=want_probs)
        res = []
        for n in names:
            if n == "logits":
                res.append(out["logits_all"].cpu().numpy())
            else:
                res.append(out["cross_probs"][int(n.split(".")[1])].cpu().numpy())
        return res
