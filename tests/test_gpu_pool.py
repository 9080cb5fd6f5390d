"""Several batches in flight on one GPU (whisper_amd.parallel.EnginePool): every batch must come back exactly as a
single engine computes it alone -- tokens, counts, boundary frames bit-identical -- whichever context ran it and
whatever ran beside it; and whisperer() must give the same result through a pool as through one engine."""
import numpy as np
import pytest
import torch

from tests.helpers import MICRO_SPECIALS as SP
from tests.helpers import synth_audio
from whisper_amd import engine as E
from whisper_amd.config import WhisperDims
from whisper_amd.parallel import EnginePool
from whisper_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup():
    dims = WhisperDims("micro-pool", 128, 2, 2, 2, 1024)
    sd = synthetic_state_dict(dims, seed=5, std=0.7 / np.sqrt(128))
    sel = [(0, 1), (1, 0)]
    return dims, sd, sel


def _batches(n_batches, B):
    out = []
    for k in range(n_batches):
        audio = torch.from_numpy(np.stack([synth_audio(100 * k + i, 480000 - 16000 * i) if False else synth_audio(100 * k + i)
                                           for i in range(B)])).to(DEV)
        n_samples = torch.tensor([480000 - 16000 * ((i + k) % 3) for i in range(B)], dtype=torch.int32)
        prompts = [[SP["sot"], 825 + (k + i) % 5, SP["ts_begin"]] + [10 + k] * (i % 3) for i in range(B)]
        out.append((audio, n_samples, prompts, 24, SP["eot"], SP["sot"], SP["ts_begin"], SP["ts_end"]))
    return out


def test_pool_results_equal_single_engine():
    dims, sd, sel = _setup()
    B, n_batches = 4, 7
    batches = _batches(n_batches, B)
    eng = E.Engine(dims, sd, dtype="bf16", max_batch=B, align_heads=sel)
    try:
        want = [{k: v.cpu().numpy() for k, v in eng.transcribe_windows(*b).items()} for b in batches]
    finally:
        eng.close()
    with EnginePool(dims, sd, inflight=3, dtype="bf16", max_batch=B, align_heads=sel) as pool:
        assert pool.max_batch == B and len(pool.engines) == 3
        got = pool.map("transcribe_windows", batches)
        for w, g in zip(want, got):
            for key in ("new_tokens", "n_new", "text_ids", "n_text", "boundaries"):
                assert np.array_equal(w[key], g[key].cpu().numpy()), key
        # the synchronous call and a second round (contexts reused, caches hold the previous batch's contents)
        again = pool.transcribe_windows(*batches[2])
        assert np.array_equal(want[2]["boundaries"], again["boundaries"].cpu().numpy())
        futs = [pool.transcribe_windows_async(*b) for b in reversed(batches)]
        for w, f in zip(reversed(want), futs):
            assert np.array_equal(w["new_tokens"], f.result()["new_tokens"].cpu().numpy())
    with pytest.raises(RuntimeError):
        pool.submit("transcribe_windows", *batches[0])


def test_pool_propagates_errors():
    dims, sd, sel = _setup()
    with EnginePool(dims, sd, inflight=2, dtype="bf16", max_batch=2, align_heads=sel) as pool:
        b = _batches(1, 4)[0]                     # four windows into contexts built for two
        with pytest.raises(Exception):
            pool.transcribe_windows(*b)
        ok = _batches(1, 2)[0]
        assert int(pool.transcribe_windows(*ok)["n_new"].shape[0]) == 2      # the pool is still usable
