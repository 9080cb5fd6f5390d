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


def test_cache_directory_lookup_follows_the_reference(tmp_path, monkeypatch):
    """get_onnx_model_paths (download.py:217-276) without the download: ~/.cache/urro_whisper/models/<name>/, aliases
    resolved, '/' -> '--', decoder_model_merged.onnx preferred, user paths only when BOTH exist."""
    from whisper_amd.model.download import cache_model_dir, get_onnx_model_paths
    monkeypatch.setenv("HOME", str(tmp_path))
    root = tmp_path / ".cache" / "urro_whisper" / "models"
    assert cache_model_dir("Turbo") == str(root / "large-v3-turbo")
    assert cache_model_dir("someone/My-Model") == str(root / "someone--my-model")
    with pytest.raises(FileNotFoundError, match="not found in cache"):
        get_onnx_model_paths("micro")
    d = root / "micro"
    d.mkdir(parents=True)
    (d / "encoder_model.onnx").write_bytes(b"")
    with pytest.raises(FileNotFoundError):
        get_onnx_model_paths("micro")                               # no decoder yet
    (d / "decoder_model.onnx").write_bytes(b"")
    assert get_onnx_model_paths("micro") == (str(d / "encoder_model.onnx"), str(d / "decoder_model.onnx"))
    (d / "decoder_model_merged.onnx").write_bytes(b"")
    assert get_onnx_model_paths("micro")[1] == str(d / "decoder_model_merged.onnx")
    # a user path that does not exist sends the lookup to the cache; two existing ones win over it
    assert get_onnx_model_paths("micro", str(tmp_path / "nope.onnx"), str(d / "decoder_model.onnx"))[0] == str(d / "encoder_model.onnx")
    ue, ud = tmp_path / "e.onnx", tmp_path / "d.onnx"
    ue.write_bytes(b""); ud.write_bytes(b"")
    assert get_onnx_model_paths("micro", str(ue), str(ud)) == (str(ue), str(ud))


def test_resolve_weights_reads_cached_files(tmp_path, monkeypatch):
    monkeypatch.setenv("HOME", str(tmp_path))
    dims = dims_for("micro")
    sd = synthetic_state_dict(dims, seed=8)
    enc, dec = _split(sd)
    d = tmp_path / ".cache" / "urro_whisper" / "models" / "micro"
    d.mkdir(parents=True)
    write_model(str(d / "encoder_model.onnx"), enc, "encoder")
    write_model(str(d / "decoder_model_merged.onnx"), dec, "decoder")
    name, got = resolve_weights("micro")
    assert name == "micro" and torch.equal(got["model.encoder.conv1.weight"], sd["model.encoder.conv1.weight"])
    with pytest.raises(FileNotFoundError, match="not found in cache"):
        resolve_weights("tiny")
