"""uw_dtw_static: the DTW entry point that allocates nothing (direction matrix in a fixed device array inside the
library) against uw_dtw with a caller workspace -- paths, lengths and boundaries bit-identical, on two calls in a
row, the second with a smaller input (the array keeps the first call's contents), plus the size limit."""
import ctypes as C

import numpy as np
import pytest
import torch

from whisper_amd import engine as E
from whisper_amd import lib as L

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dtw_static(cost, n_tok):
    lib = L.load()
    B, N, S = cost.shape
    bnd = torch.zeros((B, N + 1), dtype=torch.int32, device=DEV)
    pi = torch.zeros((B, N + S), dtype=torch.int32, device=DEV)
    pj = torch.zeros((B, N + S), dtype=torch.int32, device=DEV)
    plen = torch.zeros((B,), dtype=torch.int32, device=DEV)
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    L.check(lib.uw_dtw_static(L.ptr(cost), L.ptr(n_tok), B, N, S, L.ptr(bnd), L.ptr(pi), L.ptr(pj), L.ptr(plen), L.stream_handle()))
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    return bnd, pi, pj, plen


def _same(a, b):
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_dtw_static_matches_dtw_on_two_calls_in_a_row():
    rng = np.random.default_rng(11)
    S = 1500
    # first call: 24 windows x 300 rows, ragged lengths with exact ties; second call: smaller in every dimension
    for B, N, lens in ((24, 300, None), (5, 70, [70, 1, 33, 0, 64])):
        c = -rng.random((B, N, S))
        c[B // 2] = np.round(c[B // 2] * 4) / 4
        n = np.array(lens if lens is not None else rng.integers(1, N + 1, size=B), dtype=np.int32)
        cost = torch.from_numpy(c).to(DEV)
        n_tok = torch.from_numpy(n).to(DEV)
        ref = E.dtw(cost, n_tok)
        got = _dtw_static(cost, n_tok)
        _same(got, ref)
        assert (got[3].cpu().numpy()[n > 0] > 0).all()


def test_dtw_static_rejects_sizes_beyond_its_array():
    lib = L.load()
    B, N, S = 128, 448, 1500                      # 21.5 MB of direction matrix > the 16 MiB array
    assert B * N * ((S + 15) // 16) * 4 > 16 << 20
    rc = lib.uw_dtw_static(None, None, B, N, S, None, None, None, None, None, None)
    assert rc != 0 and b"uw_dtw_static" in lib.uw_last_error() and b"fixed device array" in lib.uw_last_error()
