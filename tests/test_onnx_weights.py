"""ONNX initialiser reader (urro_whisper/model/download.py:217-276 file contract) against files written by
tests/onnx_writer.py.  Parity unpinned: no exported Whisper graph exists offline to read."""
import numpy as np
import pytest
import torch

from tests.onnx_writer import write_model
from whisper_amd.config import dims_for
from whisper_amd.model.load import resolve_weights
from whisper_amd.model.onnx_weights import load_onnx_pair, read_onnx_graph
from whisper_amd.weights import check_state_dict, synthetic_state_dict, tensor_plan


def _split(sd):
    enc = {k: v.numpy() for k, v in sd.items() if k.startswith("model.encoder.")}
    dec = {k: v.numpy() for k, v in sd.items() if k.startswith("model.decoder.")}
    return enc, dec


@pytest.mark.parametrize("style,external,merged", [("torch", False, False), ("torch", True, True), ("short", False, True)])
def test_reader_recovers_every_weight(tmp_path, style, external, merged):
    dims = dims_for("micro")
    sd = synthetic_state_dict(dims, seed=5)
    enc, dec = _split(sd)
    if style == "short":
        enc.pop("model.encoder.embed_positions.weight")       # constant-folded: the loader regenerates the sinusoids
    ep, dp = str(tmp_path / "encoder_model.onnx"), str(tmp_path / ("decoder_model_merged.onnx" if merged else "decoder_model.onnx"))
    write_model(ep, enc, "encoder", style, external, False)
    write_model(dp, dec, "decoder", style, external, merged)
    got = load_onnx_pair(ep, dp)
    check_state_dict(dims, got)
    for name, _, _ in tensor_plan(dims):
        assert torch.equal(got[name], sd[name]), name
    inits, matmuls = read_onnx_graph(dp)
    assert any(n.endswith("/fc1/MatMul") for n, _ in matmuls) and "onnx::MatMul_999" in inits


def test_half_precision_initialisers(tmp_path):
    dims = dims_for("micro")
    sd = synthetic_state_dict(dims, seed=6)
    enc, dec = _split(sd)
    ep, dp = str(tmp_path / "e.onnx"), str(tmp_path / "d.onnx")
    write_model(ep, enc, "encoder", dtype="f16")
    write_model(dp, dec, "decoder", dtype="bf16")
    got = load_onnx_pair(ep, dp)
    k = "model.encoder.layers.0.fc1.weight"
    assert got[k].dtype == torch.float32 and torch.equal(got[k], sd[k].half().float())
    k = "model.decoder.layers.1.encoder_attn.q_proj.weight"
    ref = torch.from_numpy(((sd[k].numpy().view(np.uint32) >> 16) << 16).view(np.float32))
    assert torch.equal(got[k], ref)


def test_resolve_weights_accepts_onnx_paths(tmp_path):
    dims = dims_for("micro")
    sd = synthetic_state_dict(dims, seed=7)
    enc, dec = _split(sd)
    ep, dp = str(tmp_path / "encoder_model.onnx"), str(tmp_path / "decoder_model.onnx")
    write_model(ep, enc, "encoder")
    write_model(dp, dec, "decoder")
    name, got = resolve_weights("micro", ep, dp)
    assert name == "micro" and torch.equal(got["model.decoder.embed_tokens.weight"], sd["model.decoder.embed_tokens.weight"])
    with pytest.raises(ValueError):
        resolve_weights("tiny", ep, dp)                        # files of another shape than the named model
