"""GPU tests of the failure paths: a non-finite logits row stops its window with the tokens before it kept and leaves
the other windows of the batch alone (the reference's generation-loop failure, urro_whisper/core.py:934-980; numpy's
argmax would return the NaN's index and the reference would decode garbage); engines that whisperer() builds itself are
given back when the call ends."""
import numpy as np
import pytest
import torch

from oracle import whisper_oracle as O
from tests.helpers import MICRO_SPECIALS, synth_audio
from tests.test_gpu_api import MicroTokenizer
from whisper_amd import config as CFG
from whisper_amd import engine as E
from whisper_amd import whisperer
from whisper_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HEADS = [(0, 1), (1, 0), (1, 1)]


@pytest.fixture(scope="module")
def setup():
    dims = CFG.dims_for("micro")
    sd = synthetic_state_dict(dims, seed=0, std=0.06)
    om = O.OracleModel(sd, dims.n_heads)
    mel = np.concatenate([O.log_mel_segment(synth_audio(s), dims.n_mels) for s in (1, 2)], 0)
    enc = np.concatenate([om.encode(mel[i: i + 1]) for i in range(2)], 0).astype(np.float32)
    return dims, sd, enc


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_a_nan_window_stops_alone(setup, mode):
    dims, sd, enc = setup
    sp = MICRO_SPECIALS
    prompts = [[824, 825, 826, 924], [824, 826, 924, 9]]
    eng = E.Engine(dims, sd, dtype=mode, max_batch=2, align_heads=HEADS)
    try:
        eng.cross_kv(2, torch.from_numpy(enc).to(DEV))
        clean = eng.decode_greedy(prompts, 4 + 12, -1, sp["sot"], sp["ts_begin"], sp["ts_end"])
        assert clean["status"].tolist() == [0, 0] and clean["n_new"].tolist() == [12, 12]
        bad = enc.copy()
        bad[1, 700:710] = np.nan                                   # window 1's encoder states: every logits row is NaN
        eng.cross_kv(2, torch.from_numpy(bad).to(DEV))
        res = eng.decode_greedy(prompts, 4 + 12, -1, sp["sot"], sp["ts_begin"], sp["ts_end"])
        assert res["status"].tolist() == [0, 1]
        assert res["n_new"].tolist() == [12, 0] and res["n_text"].tolist()[1] == 0
        assert torch.equal(res["new_tokens"][0], clean["new_tokens"][0])      # the healthy window is untouched
    finally:
        eng.close()


def test_tokens_before_a_nan_step_are_kept(setup):
    """Position 6 of the decoder's position table is NaN: the step that feeds the token at position 6 produces NaN
    logits, so a window with a 4-token prompt keeps the three tokens generated before it (positions 4, 5, 6)."""
    dims, sd, enc = setup
    sp = MICRO_SPECIALS
    prompts = [[824, 825, 826, 924], [824, 826, 924, 9]]
    eng = E.Engine(dims, sd, dtype="f32", max_batch=2, align_heads=HEADS)
    try:
        eng.cross_kv(2, torch.from_numpy(enc).to(DEV))
        clean = eng.decode_greedy(prompts, 4 + 10, -1, sp["sot"], sp["ts_begin"], sp["ts_end"])
    finally:
        eng.close()
    sd_bad = dict(sd)
    pos = sd["model.decoder.embed_positions.weight"].clone()
    pos[6] = float("nan")
    sd_bad["model.decoder.embed_positions.weight"] = pos
    eng = E.Engine(dims, sd_bad, dtype="f32", max_batch=2, align_heads=HEADS)
    try:
        eng.cross_kv(2, torch.from_numpy(enc).to(DEV))
        res = eng.decode_greedy(prompts, 4 + 10, -1, sp["sot"], sp["ts_begin"], sp["ts_end"])
        assert res["status"].tolist() == [1, 1] and res["n_new"].tolist() == [3, 3]
        assert torch.equal(res["new_tokens"][:, :3], clean["new_tokens"][:, :3])
        # the whole path reports the same and marks the windows' alignment as failed (NaN rows leave DTW without a path)
        audio = torch.from_numpy(np.stack([synth_audio(1), synth_audio(2)])).to(DEV)
        n = torch.full((2,), 480000, dtype=torch.int32)
        full = eng.transcribe_windows(audio, n, prompts, 4 + 10, -1, sp["sot"], sp["ts_begin"], sp["ts_end"])
        assert full["status"].tolist() == [1, 1] and full["n_new"].tolist() == [3, 3]
    finally:
        eng.close()


def test_whisperer_gives_back_the_engines_it_builds():
    """whisperer(inflight=2) builds a pool of two library contexts, and stream=True an engine that lives as long as the
    generator: repeated calls must not keep them (a pool that is never closed keeps its contexts, worker threads and
    their HBM for ever).  The driver's free memory may wander by runtime bookkeeping (streams, code objects); a kept
    context would cost its whole size per call."""
    import gc
    dims = CFG.dims_for("micro")
    sd = synthetic_state_dict(dims, seed=0)
    probe = E.Engine(dims, sd, dtype="f32", max_batch=2, align_heads=HEADS)
    ctx_bytes = probe.device_bytes
    probe.close()
    tok = MicroTokenizer()
    audio = np.concatenate([synth_audio(5), synth_audio(6, 200000)])

    def settle():
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        return torch.cuda.mem_get_info()[0]

    kw = dict(max_tokens=16, tokenizer=tok, dtype="f32", max_batch=2)
    first = whisperer("synthetic:micro", audio, inflight=2, **kw)
    assert list(whisperer("synthetic:micro", audio, stream=True, **kw))
    free0 = settle()
    n_calls = 4
    trace = []
    for _ in range(n_calls):
        again = whisperer("synthetic:micro", audio, inflight=2, **kw)
        assert again["tokens"] == first["tokens"]
        assert list(whisperer("synthetic:micro", audio, stream=True, **kw))
        trace.append(round((free0 - settle()) / 2**20, 1))
    lost = free0 - settle()
    print(f"context {ctx_bytes >> 20} MiB; free memory changed by {lost / 2**20:.1f} MiB over {n_calls} x (pool of 2 + stream engine): {trace}")
    # 4 x 3 contexts were built and must be gone: allow less than ONE context's worth of drift in total
    assert lost < ctx_bytes, (lost, ctx_bytes)
