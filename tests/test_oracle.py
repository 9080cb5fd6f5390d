"""CPU suite: pins the oracle against the committed golden vectors (reference align.py run under stubs,
HF Whisper for the model math) and checks its DTW restatement against brute force and the C port."""
import ctypes
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import whisper_oracle as O
from tests.helpers import synth_attention
from whisper_amd import config as CFG
from whisper_amd.weights import synthetic_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_alignment_heads_match_reference_masks(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "ref_alignment_heads.json")))
    assert set(ref) == set(CFG.ALIGNMENT_HEADS)
    for k, v in ref.items():
        assert CFG.ALIGNMENT_HEADS[k] == (v["size"], v["flat"]), k


def test_alignment_heads_key_rules():
    # align.py:572-576
    assert CFG.alignment_heads_key("small", 768, "en") == "small"          # model itself is a key: overrides .en
    assert CFG.alignment_heads_key("whisper-small", 768, "en") == "small.en"
    assert CFG.alignment_heads_key("whisper-small", 768, "de") == "small"
    assert CFG.alignment_heads_key("turbo", 1280, "en") == "turbo"
    assert CFG.alignment_heads("large", 4, 20) is None                     # size mismatch -> no alignment
    assert CFG.alignment_heads("small", 12, 12)[0] == (5, 3)


def test_cost_matrix_matches_reference_align(golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_align_cost.npz"))
    for ci, key in enumerate(g["case_keys"]):
        seed, L, H, N = g[f"case{ci}_meta"]
        att = synth_attention(int(seed), int(L), int(H), int(N))
        heads = CFG.alignment_heads(str(key), int(L), int(H))
        cost = O.cost_matrix(att, heads)
        ref = g[f"case{ci}_cost"]
        assert cost.dtype == np.float64 and cost.shape == ref.shape
        assert np.array_equal(cost, ref), f"case {ci}: oracle cost matrix differs from the reference's"
        # the reference's own boundary code ran on the oracle's path: word i starts at boundary[i] * 0.02
        i1, i2 = O.dtw_symmetric1(cost)
        b = O.token_boundaries(i1, i2, int(N))
        assert np.allclose(g[f"case{ci}_starts"], np.round(b[:-1] * 0.02, 3))
        assert np.allclose(g[f"case{ci}_ends"], np.round(b[1:] * 0.02, 3))


def test_dtw_matches_bruteforce_and_handles_ties():
    rng = np.random.default_rng(3)
    for n, m in [(1, 1), (1, 7), (5, 1), (4, 9), (6, 6), (3, 12)]:
        for trial in range(4):
            c = -rng.random((n, m)) if trial < 2 else -np.round(rng.random((n, m)) * 3) / 3
            i1, i2 = O.dtw_symmetric1(c)
            assert (i1[0], i2[0]) == (0, 0) and (i1[-1], i2[-1]) == (n - 1, m - 1)
            steps = set(zip(np.diff(i1).tolist(), np.diff(i2).tolist()))
            assert steps <= {(1, 1), (0, 1), (1, 0)}
            total = c[i1, i2].sum()
            assert np.isclose(total, O.dtw_bruteforce_cost(c), rtol=0, atol=1e-12)


def test_dtw_tie_order_is_diagonal_left_up():
    # all-equal costs: every candidate ties, the scan keeps the first pattern (diagonal) wherever it exists
    c = np.zeros((3, 5))
    i1, i2 = O.dtw_symmetric1(c)
    assert i1.tolist() == [0, 0, 0, 1, 2] and i2.tolist() == [0, 1, 2, 3, 4]
    # diagonal and left tie -> diagonal wins over left
    c = np.array([[0.0, 0.0], [5.0, 0.0]])
    i1, i2 = O.dtw_symmetric1(c)
    assert list(zip(i1, i2)) == [(0, 0), (1, 1)]


def test_token_boundaries_fixup():
    i1 = np.array([0, 0, 1, 1, 2]); i2 = np.array([0, 1, 2, 3, 4])
    assert O.token_boundaries(i1, i2, 3).tolist() == [0, 2, 4, 4]      # padded with its last value
    assert O.token_boundaries(np.array([0, 0]), np.array([0, 1]), 1).tolist() == [0, 0]


def _build_c_dtw():
    out = os.path.join(ROOT, "oracle", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libdtw_ref.so")
    src = os.path.join(ROOT, "oracle", "dtw_ref.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, src, "-lm"])
    return ctypes.CDLL(so)


def c_dtw(lib, cost):
    n, m = cost.shape
    pi = np.zeros(n + m, np.int32); pj = np.zeros(n + m, np.int32); bd = np.zeros(n + 1, np.int32)
    cost = np.ascontiguousarray(cost, np.float64)
    ln = lib.dtw_symmetric1_ref(cost.ctypes.data_as(ctypes.c_void_p), n, m, pi.ctypes.data_as(ctypes.c_void_p),
                                pj.ctypes.data_as(ctypes.c_void_p), bd.ctypes.data_as(ctypes.c_void_p), 1500)
    return pi[:ln], pj[:ln], bd


def test_c_dtw_equals_numpy_dtw():
    lib = _build_c_dtw()
    rng = np.random.default_rng(11)
    for n, m, ties in [(1, 50, False), (7, 60, False), (20, 150, True), (33, 1500, False)]:
        c = -rng.random((n, m))
        if ties:
            c = np.round(c * 4) / 4
        i1, i2 = O.dtw_symmetric1(c)
        pi, pj, bd = c_dtw(lib, c)
        assert np.array_equal(pi, i1) and np.array_equal(pj, i2)
        assert np.array_equal(bd, O.token_boundaries(i1, i2, n))


def test_window_planner_and_merge_golden(golden_dir):
    from whisper_amd import align as A
    g = json.load(open(os.path.join(golden_dir, "ref_align_windows.json")))
    for plan in g["plans"]:
        got = A.calculate_sliding_windows_for_alignment(plan["total_samples"], list(range(plan["n_tokens"])))
        assert [list(w) for w in got] == plan["windows"]
    for m in g["merges"]:
        got = A.merge_overlapping_alignment_results(m["window_results"], [tuple(i) for i in m["window_info"]])
        assert got == m["merged"]


def test_model_math_matches_hf_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "hf_micro.npz"))
    dims = CFG.dims_for("micro")
    sd = synthetic_state_dict(dims, 0)
    rng = np.random.default_rng(int(g["audio_seed"]))
    audio = (0.1 * rng.standard_normal(480000)).astype(np.float32)
    mel = O.log_mel_segment(audio, 80)
    assert mel.shape == (1, 80, 3000)
    assert np.abs(mel[0][:, ::37] - g["mel_sub"]).max() < 2e-5
    assert abs(float(mel.max()) - float(g["mel_max"])) < 2e-5
    m = O.OracleModel(sd, dims.n_heads)
    enc = m.encode(mel)
    assert np.abs(enc[0, ::50] - g["enc_sub"]).max() < 2e-5
    logits, cross = m.decode_full(g["ids"].tolist(), enc, want_attn=True)
    assert np.abs(logits[0] - g["logits"]).max() < 2e-5
    assert np.abs(np.stack([c[0][:, :, ::25] for c in cross]) - g["cross_sub"]).max() < 1e-6


def test_teacher_forced_rows_equal_per_token_replay():
    dims = CFG.dims_for("micro")
    sd = synthetic_state_dict(dims, 0)
    m = O.OracleModel(sd, dims.n_heads)
    rng = np.random.default_rng(5)
    enc = rng.standard_normal((1, 1500, dims.d_model)).astype(np.float32)
    prompt, text = [824, 825, 826, 924], [5, 17, 300, 9]
    a = O.attention_rows_reference_schedule(m, enc, prompt, text)
    b = O.attention_rows_teacher_forced(m, enc, prompt, text)
    for x, y in zip(a, b):
        assert x.shape == y.shape == (1, dims.n_heads, 4, 1500)
        assert np.abs(x - y).max() < 1e-6


def test_mel_floor_uses_frames_beyond_3000():
    # a loud burst after 30 s is never encoded but lifts the max - 8 floor (audio/load.py:83-85)
    quiet = (1e-4 * np.random.default_rng(0).standard_normal(480000 + 32000)).astype(np.float32)
    loud = quiet.copy(); loud[490000:500000] += 0.5
    a, b = O.log_mel_segment(quiet, 80), O.log_mel_segment(loud, 80)
    assert a.shape == b.shape == (1, 80, 3000)
    assert b.min() > a.min() + 0.1


def test_cached_decoder_equals_cacheless_pass():
    """The KV-cached schedule (used only for the CPU baseline's "better schedule" figure) gives the cache-less pass's
    logits and last-row cross-attention."""
    from whisper_amd.config import dims_for
    from whisper_amd.weights import synthetic_state_dict
    dims = dims_for("micro")
    sd = synthetic_state_dict(dims, seed=0, std=0.06)
    om = O.OracleModel(sd, dims.n_heads)
    enc = np.random.default_rng(0).standard_normal((1, 1500, dims.d_model)).astype(np.float32)
    ids = [824, 825, 826, 924, 5, 17, 900, 3]
    ref, cross = om.decode_full(ids, enc, want_attn=True)
    dec = O.CachedDecoder(om, enc)
    lg, _ = dec.forward(ids[:5])
    assert np.abs(lg - ref[0, :5]).max() < 1e-5
    for t in range(5, len(ids)):
        lg, cw = dec.forward([ids[t]])
        assert np.abs(lg[0] - ref[0, t]).max() < 1e-5
        for l in range(dims.dec_layers):
            assert np.abs(cw[l][:, 0] - cross[l][0, :, t]).max() < 1e-6
