"""The hot path on the width / head-count / mel-bin combinations of every Whisper family the reference loads
(tiny 384/6, base 512/8, small 768/12, medium 1024/16, large-v3(-turbo) 1280/20 with 128 mel bins), with two
layers per stack and a small odd vocabulary so the CPU oracle finishes in seconds.  fp32 mode: encoder within
3e-4, greedy tokens identical to the reference's cache-less schedule, attention rows within 1e-5; bf16 mode:
teacher-forced logits within 1e-2 of the oracle evaluated in fp32 on the engine's bf16 weights."""
import numpy as np
import pytest
import torch

from oracle import whisper_oracle as O
from tests.helpers import bf16_weights, synth_audio
from whisper_amd import engine as E
from whisper_amd.config import WhisperDims
from whisper_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
VOCAB = 2049                      # odd: the logits GEMM and the argmax see a ragged last tile, as with 51865
SP = dict(sot=1800, eot=1799, ts_begin=1900, ts_end=2048)
SHAPES = [("tiny", 384, 6, 80), ("base", 512, 8, 80), ("small", 768, 12, 80), ("medium", 1024, 16, 80), ("large-v3", 1280, 20, 128)]


@pytest.mark.parametrize("name,d,heads,n_mels", SHAPES)
def test_family_shape_matches_oracle(name, d, heads, n_mels):
    dims = WhisperDims(f"{name}-2x2", d, 2, 2, heads, VOCAB, n_mels=n_mels)
    sd = synthetic_state_dict(dims, seed=3, std=0.7 / np.sqrt(d))
    om = O.OracleModel(sd, heads)
    sel = [(0, 1), (1, 0), (1, heads - 1)]
    audio = [synth_audio(11), synth_audio(12, 200000)]
    buf = np.zeros((2, 480000), np.float32)
    for i, a in enumerate(audio):
        buf[i, : len(a)] = a
    n_samples = torch.tensor([len(a) for a in audio], dtype=torch.int32)
    mel_ref = np.concatenate([O.log_mel_segment(a, n_mels) for a in audio], 0)
    enc_ref = np.concatenate([om.encode(mel_ref[i: i + 1]) for i in range(2)], 0)
    prompts = [[1800, 1801, 1802, 1900], [1798, 5, 6, 1800, 1801, 1802, 1900, 11]]
    max_tokens = 16

    eng = E.Engine(dims, sd, dtype="f32", max_batch=2, align_heads=sel)
    try:
        mel = eng.mel(torch.from_numpy(buf).to(DEV), n_samples).cpu().numpy()
        assert mel.shape == (2, n_mels, 3000) and np.abs(mel - mel_ref).max() < 5e-5
        enc = eng.encode(torch.from_numpy(mel_ref).to(DEV)).cpu().numpy()
        assert np.abs(enc - enc_ref).max() < 3e-4, np.abs(enc - enc_ref).max()
        res = eng.transcribe_windows(torch.from_numpy(buf).to(DEV), n_samples, prompts, max_tokens, SP["eot"], SP["sot"],
                                     SP["ts_begin"], SP["ts_end"])
        for b, pr in enumerate(prompts):
            new = O.greedy_reference_schedule(om, enc_ref[b: b + 1], pr, max_tokens, SP["eot"])
            text = O.filter_text_tokens(new, SP["sot"], SP["ts_begin"], SP["ts_end"])
            nn, nt = int(res["n_new"][b]), int(res["n_text"][b])
            assert res["new_tokens"][b, :nn].cpu().tolist() == new
            assert res["text_ids"][b, :nt].cpu().tolist() == text
            if text:
                rows_ref = O.attention_rows_reference_schedule(om, enc_ref[b: b + 1], pr, text)
                rows = eng.attn_rows(2, nt)[b].cpu().numpy()
                assert np.abs(rows - np.stack([rows_ref[l][0, h] for l, h in sel])).max() < 1e-5
                bd = res["boundaries"][b, : nt + 1].cpu().numpy()
                assert bd[0] == 0 and np.all(np.diff(bd) >= 0) and bd[-1] <= 1500
    finally:
        eng.close()

    eng = E.Engine(dims, sd, dtype="bf16", max_batch=2, align_heads=sel)
    try:
        eng.cross_kv(1, torch.from_numpy(enc_ref[:1]).to(DEV))
        seq = prompts[0] + O.greedy_reference_schedule(om, enc_ref[:1], prompts[0], 14, eot=-1)
        ref = O.OracleModel(bf16_weights(sd), heads).decode_full(seq, enc_ref[:1])[0]
        got = eng.decoder_forward(torch.tensor([seq], dtype=torch.int32), logits_all=True)["logits_all"][0].cpu().numpy()
        assert np.abs(got - ref).max() < 1e-2, np.abs(got - ref).max()
        res = eng.transcribe_windows(torch.from_numpy(buf).to(DEV), n_samples, prompts, max_tokens, SP["eot"], SP["sot"],
                                     SP["ts_begin"], SP["ts_end"])
        assert int(res["n_new"][0]) == max_tokens - len(prompts[0])
    finally:
        eng.close()
