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
    call = lambda: L.check(lib.uw_dtw_static(L.ptr(cost), L.ptr(n_tok), B, N, S, L.ptr(bnd), L.ptr(pi), L.ptr(pj), L.ptr(plen),
                                             L.stream_handle()))
    call()      # the first call on a device may load the code object and resolve the array's address
    torch.cuda.synchronize()
    # free bytes as the DRIVER counts them (hipMemGetInfo): this also sees a hipMalloc made inside the library, which
    # torch's own allocator statistics would not
    free_before = torch.cuda.mem_get_info()[0]
    torch_before = torch.cuda.memory_allocated()
    call()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before and torch.cuda.memory_allocated() == torch_before
    return bnd, pi, pj, plen


def _oracle_window(cost_row, n):
    """One window through the C restatement of dtw-python's recursion (oracle/dtw_ref.c)."""
    import os
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_build", "libdtw_ref.so")
    ref = C.CDLL(so)
    c = np.ascontiguousarray(cost_row[:n])
    m = c.shape[1]
    pi, pj, bd = np.zeros(n + m, np.int32), np.zeros(n + m, np.int32), np.zeros(n + 1, np.int32)
    ln = ref.dtw_symmetric1_ref(c.ctypes.data_as(C.c_void_p), n, m, pi.ctypes.data_as(C.c_void_p), pj.ctypes.data_as(C.c_void_p),
                                bd.ctypes.data_as(C.c_void_p), 1500)
    return ln, pi[:ln], pj[:ln], bd


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
        # ... and against the oracle directly: the window with exact ties and the longest one
        for b in {B // 2, int(np.argmax(n))}:
            if n[b] == 0:
                continue
            ln, oi, oj, obd = _oracle_window(c[b], int(n[b]))
            assert ln == int(got[3][b]) and np.array_equal(got[1][b, :ln].cpu().numpy(), oi) and np.array_equal(got[2][b, :ln].cpu().numpy(), oj)
            assert np.array_equal(got[0][b, : n[b] + 1].cpu().numpy(), obd)


def test_dtw_static_rejects_sizes_beyond_its_array():
    lib = L.load()
    B, N, S = 128, 448, 1500                      # 21.5 MB of direction matrix > the 16 MiB array
    assert B * N * ((S + 15) // 16) * 4 > 16 << 20
    rc = lib.uw_dtw_static(None, None, B, N, S, None, None, None, None, None)
    assert rc != 0 and b"uw_dtw_static" in lib.uw_last_error() and b"fixed device array" in lib.uw_last_error()
