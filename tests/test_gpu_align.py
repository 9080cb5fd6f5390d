"""GPU parity: cost-matrix and DTW kernels against the oracle, through the C ABI (whisper_amd.engine)."""
import os

import numpy as np
import pytest
import torch

from oracle import whisper_oracle as O
from tests.helpers import synth_attention
from tests.test_oracle import _build_c_dtw, c_dtw
from whisper_amd import config as CFG
from whisper_amd import engine as E

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dtw_gpu(cost, n_tok):
    bnd, pi, pj, plen = E.dtw(torch.from_numpy(np.ascontiguousarray(cost)).to(DEV), torch.from_numpy(n_tok))
    torch.cuda.synchronize()
    return bnd.cpu().numpy(), pi.cpu().numpy(), pj.cpu().numpy(), plen.cpu().numpy()


def _assert_paths_equal(cost, n_tok):
    bnd, pi, pj, plen = _dtw_gpu(cost, n_tok)
    for b in range(cost.shape[0]):
        n = int(n_tok[b])
        if n == 0:
            assert plen[b] == 0
            continue
        i1, i2 = O.dtw_symmetric1(cost[b, :n])
        L = int(plen[b])
        assert L == len(i1), (b, L, len(i1))
        assert np.array_equal(pi[b, :L], i1) and np.array_equal(pj[b, :L], i2), f"window {b}: path differs"
        assert np.array_equal(bnd[b, : n + 1], O.token_boundaries(i1, i2, n))


def test_cost_matrix_matches_reference_golden(golden_dir):
    """rows -> cost against the matrices the reference's own align.py produced (tolerance: fp32 softmax
    evaluated with a different exp and summation order, 2e-6 relative)."""
    g = np.load(os.path.join(golden_dir, "ref_align_cost.npz"))
    for ci, key in enumerate(g["case_keys"]):
        seed, L, H, N = (int(x) for x in g[f"case{ci}_meta"])
        att = synth_attention(seed, L, H, N)
        heads = CFG.alignment_heads(str(key), L, H)
        rows = np.stack([att[l][0, h] for l, h in heads])[None]          # [1, n_sel, N, S]
        cost = E.cost_matrix(torch.from_numpy(rows).to(DEV), torch.tensor([N], dtype=torch.int32)).cpu().numpy()[0]
        ref = g[f"case{ci}_cost"]
        assert np.allclose(cost, ref, rtol=2e-6, atol=0), np.abs(cost / ref - 1).max()


def test_cost_matrix_ragged_batch_and_median_edges():
    rng = np.random.default_rng(0)
    B, n_sel, N, S = 3, 5, 40, 1500
    rows = rng.random((B, n_sel, N, S), dtype=np.float32)
    rows[:, :, :, :4] *= 50.0      # stress the reflected edges of the median window
    rows[:, :, :, -4:] *= 50.0
    n_tok = np.array([40, 1, 17], dtype=np.int32)
    cost = E.cost_matrix(torch.from_numpy(rows).to(DEV), torch.from_numpy(n_tok)).cpu().numpy()
    for b in range(B):
        n = int(n_tok[b])
        ref = O.cost_matrix([rows[b, k, :n][None, None] for k in range(n_sel)], [(k, 0) for k in range(n_sel)])
        assert np.allclose(cost[b, :n], ref, rtol=3e-5, atol=1e-30)   # exp() of arguments down to -50: |x| * eps
        assert np.all(cost[b, n:] == 0)    # rows past n_tok are not touched


def test_dtw_bit_exact_random_and_ties():
    rng = np.random.default_rng(1)
    S = 1500
    cost = -rng.random((4, 200, S))
    cost[1] = np.round(cost[1] * 4) / 4          # four cost levels: ties everywhere
    cost[2] = 0.0                                # every candidate ties
    cost[3, :, :] = -np.float32(rng.random((200, S), dtype=np.float32)).astype(np.float64)   # f32-valued like the real matrix
    _assert_paths_equal(cost, np.array([200, 130, 65, 64], dtype=np.int32))


def test_dtw_edge_sizes():
    rng = np.random.default_rng(2)
    S = 1500
    cost = -rng.random((5, 448, S))
    _assert_paths_equal(cost, np.array([448, 1, 2, 63, 0], dtype=np.int32))     # max rows, single row, band edges, empty
    cost = -rng.random((2, 130, S))
    _assert_paths_equal(cost, np.array([129, 128], dtype=np.int32))


def test_dtw_on_reference_golden_costs(golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_align_cost.npz"))
    for ci in range(len(g["case_keys"])):
        ref = g[f"case{ci}_cost"]
        n = ref.shape[0]
        bnd, pi, pj, plen = _dtw_gpu(ref[None], np.array([n], dtype=np.int32))
        # the reference's own boundary code turned the path into these word times (one word per token)
        assert np.allclose(np.round(bnd[0, :n] * 0.02, 3), g[f"case{ci}_starts"])
        assert np.allclose(np.round(bnd[0, 1: n + 1] * 0.02, 3), g[f"case{ci}_ends"])


def test_dtw_full_batch_properties_and_c_oracle():
    """Bench-sized batch (128 windows x 128 tokens): structural properties everywhere, C oracle on a sample."""
    rng = np.random.default_rng(3)
    B, N, S = 128, 128, 1500
    cost = -rng.random((B, N, S), dtype=np.float32).astype(np.float64)
    n_tok = rng.integers(1, N + 1, size=B).astype(np.int32)
    bnd, pi, pj, plen = _dtw_gpu(cost, n_tok)
    lib = _build_c_dtw()
    for b in range(B):
        n, L = int(n_tok[b]), int(plen[b])
        i1, i2 = pi[b, :L], pj[b, :L]
        assert (i1[0], i2[0]) == (0, 0) and (i1[-1], i2[-1]) == (n - 1, S - 1)
        di, dj = np.diff(i1), np.diff(i2)
        assert np.all((di >= 0) & (dj >= 0) & (di + dj >= 1) & (di <= 1) & (dj <= 1))
        assert np.all(np.diff(bnd[b, : n + 1]) >= 0) and bnd[b, 0] == 0
        if b % 16 == 0:
            ci, cj, cb = c_dtw(lib, cost[b, :n])
            assert np.array_equal(ci, i1) and np.array_equal(cj, i2) and np.array_equal(cb, bnd[b, : n + 1])


def test_dtw_without_a_path_is_reported_per_window():
    """A NaN cost row leaves cells without a predecessor: dtw-python raises and the reference returns no timestamps
    for that chunk (align.py:823-827).  Here the window is marked (boundaries[b][0] = -1, path_len < 0) and the other
    windows of the batch are unaffected."""
    rng = np.random.default_rng(3)
    N, S = 20, 1500
    cost = -rng.random((3, N, S))
    bad = cost.copy()
    bad[1, 7, :] = np.nan
    n_tok = torch.tensor([N, N, N], dtype=torch.int32)
    bnd_ok, _, _, len_ok = E.dtw(torch.from_numpy(cost).cuda(), n_tok)
    bnd, _, _, plen = E.dtw(torch.from_numpy(bad).cuda(), n_tok)
    bnd, plen, bnd_ok = bnd.cpu().numpy(), plen.cpu().numpy(), bnd_ok.cpu().numpy()
    assert (len_ok.cpu().numpy() > 0).all()
    assert plen[1] < 0 and bnd[1, 0] == -1
    for b in (0, 2):
        assert plen[b] > 0 and np.array_equal(bnd[b], bnd_ok[b])
