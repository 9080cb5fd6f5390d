"""The benchmark configuration at full size (whisper-small shapes, 128 windows): properties that do not need the
CPU oracle at that size — duplicated windows give bit-identical results wherever they sit in the batch, attention
rows are probability rows, boundaries are monotone frames — plus the oracle on ONE window of the real 12 + 12
layer shape in fp32 (tokens identical to the reference's cache-less schedule for the first steps)."""
import numpy as np
import pytest
import torch

from oracle import whisper_oracle as O
from tests.helpers import WHISPER_SPECIALS as SP
from tests.helpers import synth_audio
from whisper_amd import engine as E
from whisper_amd.config import alignment_heads, alignment_heads_key, dims_for
from whisper_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def small():
    dims = dims_for("small")
    sd = synthetic_state_dict(dims, seed=0)
    heads = alignment_heads(alignment_heads_key("small", dims.d_model, "en"), dims.dec_layers, dims.n_heads)
    return dims, sd, heads


def test_full_batch_invariants_bf16(small):
    dims, sd, heads = small
    B = 128
    eng = E.Engine(dims, sd, dtype="bf16", max_batch=B, align_heads=heads)
    try:
        base = [synth_audio(40 + i) for i in range(4)]
        order = [i % 4 for i in range(B)]                      # every window appears 32 times, spread over the batch
        audio = torch.from_numpy(np.stack([base[i] for i in order])).to(DEV)
        n_samples = torch.full((B,), 480000, dtype=torch.int32)
        prompt = [SP["sot"], 50259, 50359, SP["ts_begin"]]
        prompts = [prompt + [1000 + i] for i in order]
        res = eng.transcribe_windows(audio, n_samples, prompts, len(prompt) + 1 + 12, SP["eot"], SP["sot"], SP["ts_begin"], SP["ts_end"])
        n_new = res["n_new"].cpu().numpy()
        n_text = res["n_text"].cpu().numpy()
        assert (n_new == 12).all()
        toks = res["new_tokens"].cpu().numpy()
        bnd = res["boundaries"].cpu().numpy()
        rows = eng.attn_rows(B, int(n_text.max())).cpu().numpy()
        for b in range(4, B):
            assert np.array_equal(toks[b], toks[b % 4]) and n_text[b] == n_text[b % 4]
            assert np.array_equal(bnd[b], bnd[b % 4])
            assert np.array_equal(rows[b, :, : n_text[b]], rows[b % 4, :, : n_text[b]])
        for b in range(4):
            nt = int(n_text[b])
            if nt:
                assert np.allclose(rows[b, :, :nt].sum(-1), 1.0, atol=2e-3)
                bd = bnd[b, : nt + 1]
                assert bd[0] == 0 and np.all(np.diff(bd) >= 0) and bd[-1] <= 1500 and bd[-1] == bd[-2]
    finally:
        eng.close()


def test_one_real_size_window_matches_oracle_fp32(small):
    dims, sd, heads = small
    om = O.OracleModel(sd, dims.n_heads)
    a = synth_audio(77)
    mel = O.log_mel_segment(a, dims.n_mels)
    enc_ref = om.encode(mel)
    prompt = [SP["sot"], 50259, 50359, SP["ts_begin"], 2425]
    max_tokens = len(prompt) + 5
    new = O.greedy_reference_schedule(om, enc_ref, prompt, max_tokens, SP["eot"])
    text = O.filter_text_tokens(new, SP["sot"], SP["ts_begin"], SP["ts_end"])
    # 44 copies of the window: 44 x 12 heads = 528 (window, head) pairs, which selects the decode-attention form
    # that full batches run (4 waves per workgroup); every copy must match the oracle
    B = 44
    eng = E.Engine(dims, sd, dtype="f32", max_batch=B, align_heads=heads)
    try:
        enc = eng.encode(torch.from_numpy(mel).to(DEV)).cpu().numpy()
        assert np.abs(enc - enc_ref).max() < 5e-4, np.abs(enc - enc_ref).max()
        res = eng.transcribe_windows(torch.from_numpy(np.repeat(a[None], B, 0)).to(DEV), torch.full((B,), 480000, dtype=torch.int32),
                                     [prompt] * B, max_tokens, SP["eot"], SP["sot"], SP["ts_begin"], SP["ts_end"])
        rows_ref = O.attention_rows_reference_schedule(om, enc_ref, prompt, text) if text else None
        for b in (0, 17, B - 1):
            nn, nt = int(res["n_new"][b]), int(res["n_text"][b])
            assert res["new_tokens"][b, :nn].cpu().tolist() == new
            assert res["text_ids"][b, :nt].cpu().tolist() == text
            if text:
                rows = eng.attn_rows(B, nt)[b].cpu().numpy()
                assert np.abs(rows - np.stack([rows_ref[l][0, h] for l, h in heads])).max() < 1e-5
    finally:
        eng.close()


@pytest.mark.parametrize("model,B", [("medium", 6), ("large-v3-turbo", 8)])
def test_largest_shapes_run_and_are_batch_invariant(model, B):
    """The two largest configurations at their real depth (24 + 24 layers at d = 1024; 32 + 4 layers at d = 1280 with
    128 mel bins), bf16: duplicated windows agree bit for bit, rows are probability rows, boundaries monotone."""
    dims = dims_for(model)
    sd = synthetic_state_dict(dims, seed=1)
    heads = alignment_heads(alignment_heads_key(model, dims.d_model, "en"), dims.dec_layers, dims.n_heads)
    eng = E.Engine(dims, sd, dtype="bf16", max_batch=B, align_heads=heads)
    del sd
    try:
        base = [synth_audio(60), synth_audio(61, 300000)]
        order = [i % 2 for i in range(B)]
        buf = np.zeros((B, 480000), np.float32)
        for b, i in enumerate(order):
            buf[b, : len(base[i])] = base[i]
        n_samples = torch.tensor([len(base[i]) for i in order], dtype=torch.int32)
        prompt = [SP["sot"], 50259, 50359, SP["ts_begin"]]
        res = eng.transcribe_windows(torch.from_numpy(buf).to(DEV), n_samples, [prompt] * B, len(prompt) + 10, SP["eot"], SP["sot"],
                                     SP["ts_begin"], SP["ts_end"])
        n_text = res["n_text"].cpu().numpy()
        assert (res["n_new"].cpu().numpy() == 10).all()
        toks, bnd = res["new_tokens"].cpu().numpy(), res["boundaries"].cpu().numpy()
        rows = eng.attn_rows(B, int(max(1, n_text.max()))).cpu().numpy()
        for b in range(2, B):
            assert np.array_equal(toks[b], toks[b % 2]) and np.array_equal(bnd[b], bnd[b % 2])
            assert np.array_equal(rows[b, :, : n_text[b]], rows[b % 2, :, : n_text[b]])
        for b in range(2):
            nt = int(n_text[b])
            if nt:
                assert np.allclose(rows[b, :, :nt].sum(-1), 1.0, atol=2e-3)
                bd = bnd[b, : nt + 1]
                assert bd[0] == 0 and np.all(np.diff(bd) >= 0) and bd[-1] <= 1500
    finally:
        eng.close()
