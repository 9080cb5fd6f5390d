"""Writes minimal ONNX files (protobuf wire format by hand) shaped like torch-exported Whisper graphs, for
testing the initialiser reader: named conv / bias / LayerNorm / embedding tensors, anonymous transposed
`onnx::MatMul_<n>` Linear weights consumed by module-named MatMul nodes, optional external data file, optional
If node whose branches hold the MatMul nodes (merged decoder)."""
import struct

import numpy as np


def _vi(x: int) -> bytes:
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        if x:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(fno: int, payload: bytes) -> bytes:
    return _vi((fno << 3) | 2) + _vi(len(payload)) + payload


def _int(fno: int, v: int) -> bytes:
    return _vi(fno << 3) + _vi(v)


def tensor_proto(name, arr, external=None, dtype="f32", packed_dims=True):
    arr = np.ascontiguousarray(arr, dtype=np.float32)
    if dtype == "f16":
        data, code = arr.astype(np.float16).tobytes(), 10
    elif dtype == "bf16":
        data, code = (arr.view(np.uint32) >> 16).astype(np.uint16).tobytes(), 16      # truncation
    else:
        data, code = arr.tobytes(), 1
    if packed_dims:
        msg = _ld(1, b"".join(_vi(d) for d in arr.shape))
    else:
        msg = b"".join(_int(1, d) for d in arr.shape)
    msg += _int(2, code) + _ld(8, name.encode())
    if external is not None:
        fh, fname = external
        off = fh.tell()
        fh.write(data)
        for k, v in (("location", fname), ("offset", str(off)), ("length", str(len(data)))):
            msg += _ld(13, _ld(1, k.encode()) + _ld(2, v.encode()))
        msg += _int(14, 1)
    else:
        msg += _ld(9, data)
    return msg


def node_proto(name, op, inputs, outputs, subgraphs=()):
    msg = b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs)
    msg += _ld(3, name.encode()) + _ld(4, op.encode())
    for attr_name, g in subgraphs:
        msg += _ld(5, _ld(1, attr_name.encode()) + _ld(6, g) + _int(20, 5))
    return msg


def write_model(path, state, role, style="torch", external=False, merged=False, dtype="f32"):
    """state: {hf_name: np.ndarray} for one stack (names starting with model.<role>.)."""
    ext = None
    fh = None
    if external:
        fh = open(path + "_data", "wb")
        ext = (fh, path.split("/")[-1] + "_data")
    inits, nodes = [], []
    k = 0
    for name, arr in state.items():
        short = name[len("model."):] if style == "short" else name
        is_linear = arr.ndim == 2 and name.endswith(".weight") and "embed_" not in name
        if is_linear:
            k += 1
            anon = f"onnx::MatMul_{1000 + k}"
            inits.append(tensor_proto(anon, arr.T, ext if (external and arr.size > 4096) else None, dtype))
            module = short[: -len(".weight")]
            if style == "short":
                module = module.split(".", 1)[1]              # drop the "encoder." / "decoder." root as well
            parts = module.split(".")
            segs, i = [], 0
            while i < len(parts):                              # "layers.0" stays one path segment
                if parts[i] == "layers":
                    segs.append(f"layers.{parts[i + 1]}")
                    i += 2
                else:
                    segs.append(parts[i])
                    i += 1
            nodes.append(node_proto("/" + "/".join(segs) + "/MatMul", "MatMul", [f"x{k}", anon], [f"y{k}"]))
        else:
            inits.append(tensor_proto(short, arr, None, dtype, packed_dims=(k % 2 == 0)))
    inits.append(tensor_proto("onnx::MatMul_999", np.zeros((4, 6), np.float32)))      # an unrelated anonymous constant
    nodes.append(node_proto("/proj_out/MatMul", "MatMul", ["h", "onnx::MatMul_999"], ["logits"]))
    if merged:
        half = len(nodes) // 2
        then_g = b"".join(_ld(1, n) for n in nodes[:half]) + _ld(2, b"then")
        else_g = b"".join(_ld(1, n) for n in nodes[half:]) + _ld(2, b"else")
        nodes = [node_proto("optimum::if", "If", ["use_cache_branch"], ["logits"], [("then_branch", then_g), ("else_branch", else_g)])]
    graph = b"".join(_ld(1, n) for n in nodes) + _ld(2, b"main_graph") + b"".join(_ld(5, t) for t in inits)
    model = _int(1, 8) + _ld(2, b"pytorch") + _ld(7, graph)
    with open(path, "wb") as out:
        out.write(model)
    if fh:
        fh.close()
