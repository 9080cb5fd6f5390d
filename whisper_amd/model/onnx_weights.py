"""Weights out of the ONNX files the reference caches (urro_whisper/model/download.py:217-276: `encoder_model.onnx`
plus `decoder_model.onnx` or `decoder_model_merged.onnx`, each with an optional sibling `<file>_data`).

Only what weight ingestion needs of the ONNX protobuf schema is decoded, by hand (no onnx package offline):
ModelProto.graph (field 7); GraphProto.node (1) and .initializer (5); NodeProto.input (1), .name (3), .op_type (4),
.attribute (5); AttributeProto.g (6) and .graphs (11) for the If-branches of a merged decoder; TensorProto.dims (1),
.data_type (2), .float_data (4), .name (8), .raw_data (9), .external_data (13: key/value pairs `location`, `offset`,
`length`), .data_location (14).

Name recovery: torch's exporter keeps parameter names for convolutions, biases, LayerNorms and embeddings but
stores each Linear weight as an anonymous, transposed `onnx::MatMul_<n>` initialiser; the MatMul node that
consumes it is named after the module (`/model/decoder/layers.0/self_attn/q_proj/MatMul`), which gives the
parameter name back.  Parity unpinned: no exported Whisper graph exists offline, the reader is tested against
files written by tests/onnx_writer.py in both naming styles.
"""
from __future__ import annotations

import mmap
import os
import struct
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch

_LEAVES = ("layers", "conv1", "conv2", "embed_tokens", "embed_positions", "layer_norm")
_DTYPES = {1: np.float32, 10: np.float16, 11: np.float64, 6: np.int32, 7: np.int64}


def _varint(buf, i: int) -> Tuple[int, int]:
    shift = result = 0
    while True:
        b = buf[i]
        i += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, i
        shift += 7


def _fields(buf) -> Iterator[Tuple[int, int, object]]:
    """(field number, wire type, value) of one message; length-delimited values are memoryview slices."""
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 2:
            ln, i = _varint(buf, i)
            v = buf[i: i + ln]
            i += ln
        elif wt == 5:
            v = buf[i: i + 4]
            i += 4
        elif wt == 1:
            v = buf[i: i + 8]
            i += 8
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, v


def _packed_varints(v) -> List[int]:
    out, i = [], 0
    while i < len(v):
        x, i = _varint(v, i)
        out.append(x)
    return out


def _tensor(buf, base_dir: str) -> Tuple[str, Optional[np.ndarray]]:
    dims: List[int] = []
    dtype = 0
    name = ""
    raw = None
    floats = None
    ext: Dict[str, str] = {}
    location = 0
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dims += _packed_varints(v) if wt == 2 else [v]
        elif fno == 2:
            dtype = v
        elif fno == 4:
            floats = np.frombuffer(v, dtype="<f4") if wt == 2 else np.append(floats if floats is not None else [], struct.unpack("<f", v))
        elif fno == 8:
            name = bytes(v).decode()
        elif fno == 9:
            raw = v
        elif fno == 13:
            kv = {f: bytes(x).decode() for f, _, x in _fields(v)}
            ext[kv.get(1, "")] = kv.get(2, "")
        elif fno == 14:
            location = v
    if dtype == 16:                                   # bfloat16: widen through the bit pattern
        np_dtype = np.uint16
    elif dtype in _DTYPES:
        np_dtype = _DTYPES[dtype]
    else:
        return name, None                             # strings, bools, ...: never weights
    if location == 1 or (raw is None and floats is None and ext):
        path = os.path.join(base_dir, ext["location"])
        off = int(ext.get("offset", "0") or 0)
        count = int(np.prod(dims)) if dims else 1
        arr = np.fromfile(path, dtype=np_dtype, count=count, offset=off)
    elif raw is not None:
        arr = np.frombuffer(raw, dtype=np_dtype)
    elif floats is not None:
        arr = np.asarray(floats, dtype=np.float32)
    else:
        return name, None
    if dtype == 16:
        arr = (arr.astype(np.uint32) << 16).view(np.float32)
    return name, np.array(arr, copy=True).reshape(dims)


def _walk_graph(buf, base_dir: str, inits: Dict[str, np.ndarray], matmuls: List[Tuple[str, str]]) -> None:
    for fno, wt, v in _fields(buf):
        if fno == 5 and wt == 2:
            name, arr = _tensor(v, base_dir)
            if arr is not None:
                inits[name] = arr
        elif fno == 1 and wt == 2:                     # NodeProto
            inputs: List[str] = []
            nname = op = ""
            for f2, w2, x in _fields(v):
                if f2 == 1:
                    inputs.append(bytes(x).decode())
                elif f2 == 3:
                    nname = bytes(x).decode()
                elif f2 == 4:
                    op = bytes(x).decode()
                elif f2 == 5:                          # AttributeProto: recurse into sub-graphs (If branches)
                    for f3, w3, y in _fields(x):
                        if f3 in (6, 11) and w3 == 2:
                            _walk_graph(y, base_dir, inits, matmuls)
            if op == "MatMul" and len(inputs) == 2:
                matmuls.append((nname, inputs[1]))


def read_onnx_graph(path: str) -> Tuple[Dict[str, np.ndarray], List[Tuple[str, str]]]:
    """-> (initialisers by name, [(MatMul node name, weight initialiser name)])."""
    inits: Dict[str, np.ndarray] = {}
    matmuls: List[Tuple[str, str]] = []
    base = os.path.dirname(os.path.abspath(path))
    with open(path, "rb") as fh:
        mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
        try:
            view = memoryview(mm)
            for fno, wt, v in _fields(view):
                if fno == 7 and wt == 2:
                    _walk_graph(v, base, inits, matmuls)
            del v, view
        finally:
            try:
                mm.close()
            except BufferError:        # a slice is still referenced by a caller's traceback; the OS reclaims it
                pass
    return inits, matmuls


def _canonical(dotted: str, default_role: str) -> Optional[str]:
    """`model.decoder.layers.0.fc1.weight` from any of `decoder.layers.0...`, `layers.0...`, `model.model.decoder...`."""
    toks = dotted.split(".")
    role = default_role
    for i, t in enumerate(toks):
        if t in ("encoder", "decoder"):
            role = t
        if t in _LEAVES:
            return f"model.{role}." + ".".join(toks[i:])
    return None


def onnx_state_dict(path: str, role: str) -> Dict[str, torch.Tensor]:
    """HF-named fp32 tensors found in one graph file.  role: "encoder" | "decoder" (used when names do not say)."""
    inits, matmuls = read_onnx_graph(path)
    sd: Dict[str, torch.Tensor] = {}
    for name, arr in inits.items():
        if name.startswith("onnx::") or arr.ndim == 0:
            continue
        canon = _canonical(name, role)
        if canon and (canon.endswith(".weight") or canon.endswith(".bias")):
            sd[canon] = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))
    for node, winit in matmuls:
        if winit not in inits or inits[winit].ndim != 2:
            continue
        parts = [p for p in node.strip("/").split("/") if p]
        if not parts or not parts[-1].startswith("MatMul"):
            continue
        canon = _canonical(".".join(parts[:-1]) + ".weight", role)
        if canon and canon not in sd and "proj_out" not in node:
            sd[canon] = torch.from_numpy(np.ascontiguousarray(inits[winit].T, dtype=np.float32))   # [in, out] -> [out, in]
    return sd


def load_onnx_pair(encoder_path: str, decoder_path: str) -> Dict[str, torch.Tensor]:
    """One state dict from the encoder file and the decoder (plain or merged) file."""
    from ..weights import sinusoids
    sd = onnx_state_dict(encoder_path, "encoder")
    sd.update({k: v for k, v in onnx_state_dict(decoder_path, "decoder").items() if k.startswith("model.decoder.")})
    key = "model.encoder.embed_positions.weight"
    if key not in sd and "model.encoder.conv1.weight" in sd:      # constant-folded away: Whisper's fixed sinusoids
        sd[key] = sinusoids(1500, sd["model.encoder.conv1.weight"].shape[0])
    return sd
