"""GPU parity of mel, encoder, decoder, greedy decode and the end-to-end window path against the oracle.

Tolerances (stated by the path's parity contract): fp32 mode logits within 1e-4 abs and greedy tokens
identical; bf16 mode is checked teacher-forced, logits within 1e-2 abs of the oracle evaluated in fp32 on the
weights the bf16 engine holds (tests.helpers.bf16_weights: rounding the weights alone moves the fp32 oracle by about
as much again, which no bf16-weight implementation can avoid); DTW bit-exact on equal cost matrices; mel within
5e-5 abs of the fp32 oracle.
"""
import numpy as np
import pytest
import torch

from oracle import whisper_oracle as O
from tests.helpers import MICRO_SPECIALS, bf16_weights, synth_audio
from whisper_amd import config as CFG
from whisper_amd import engine as E
from whisper_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HEADS = [(0, 1), (1, 0), (1, 1)]


@pytest.fixture(scope="module")
def micro():
    dims = CFG.dims_for("micro")
    sd = synthetic_state_dict(dims, seed=0, std=0.06)     # wide enough that tokens vary from step to step
    om = O.OracleModel(sd, dims.n_heads)
    om_r = O.OracleModel(bf16_weights(sd), dims.n_heads)      # fp32 arithmetic on the bf16 engine's weights
    eng32 = E.Engine(dims, sd, dtype="f32", max_batch=4, align_heads=HEADS)
    engbf = E.Engine(dims, sd, dtype="bf16", max_batch=4, align_heads=HEADS)
    audio = [synth_audio(1), synth_audio(2, 300000)]
    mel = np.concatenate([O.log_mel_segment(a, dims.n_mels) for a in audio], 0)
    enc = np.concatenate([om.encode(mel[i: i + 1]) for i in range(2)], 0)
    enc_r = np.concatenate([om_r.encode(mel[i: i + 1]) for i in range(2)], 0)
    yield dict(dims=dims, sd=sd, om=om, f32=eng32, bf16=engbf, audio=audio, mel=mel, enc=enc,
               ref={"f32": om, "bf16": om_r}, enc_ref={"f32": enc, "bf16": enc_r})
    eng32.close()
    engbf.close()


def test_mel_matches_oracle(micro):
    eng = micro["f32"]
    a0, a1 = micro["audio"]
    long = synth_audio(3, 480000 + 80000)             # 35 s segment: frames past 3000 only feed the floor
    long[490000:500000] *= 20.0
    quiet = np.zeros(3200, np.float32)                # shortest chunk the reference processes, all silence
    items = [a0, a1, long, quiet]
    stride = max(len(a) for a in items)
    buf = np.zeros((len(items), stride), np.float32)
    for i, a in enumerate(items):
        buf[i, : len(a)] = a
    buf[1, 300000:] = 7.0                             # samples past n_samples must be ignored
    n = torch.tensor([len(a) for a in items], dtype=torch.int32)
    mel = eng.mel(torch.from_numpy(buf).to(DEV), n).cpu().numpy()
    assert mel.shape == (4, 80, 3000)
    for i, a in enumerate(items):
        ref = O.log_mel_segment(a, 80)[0]
        assert np.abs(mel[i] - ref).max() < 5e-5, (i, np.abs(mel[i] - ref).max())
    assert np.abs(mel[3] + 1.5).max() < 1e-6                     # silence: log10(1e-10) = -10 -> (-10 + 4) / 4


@pytest.mark.parametrize("mode,tol", [("f32", 1e-4), ("bf16", 3e-2)])
def test_encoder_matches_oracle(micro, mode, tol):
    """Encoder states (LayerNorm outputs of unit scale; the contract's bf16 figure is for logits, where this error
    arrives averaged over the 1500 frames)."""
    enc = micro[mode].encode(torch.from_numpy(micro["mel"]).to(DEV)).cpu().numpy()
    assert enc.shape == micro["enc"].shape
    assert np.abs(enc - micro["enc_ref"][mode]).max() < tol


@pytest.mark.parametrize("mode,tol_logits,tol_probs", [("f32", 1e-4, 1e-6), ("bf16", 1e-2, 1e-4)])
def test_decoder_forward_matches_oracle(micro, mode, tol_logits, tol_probs):
    eng, om, enc = micro[mode], micro["ref"][mode], micro["enc"]
    eng.cross_kv(2, torch.from_numpy(enc).to(DEV))
    ids = [[824, 825, 826, 924, 5, 17, 900, 3, 321, 77, 12], [824, 826, 924, 9, 9, 9, 1000, 1, 2, 3, 4]]
    out = eng.decoder_forward(torch.tensor(ids, dtype=torch.int32), logits_all=True, logits_last=True, cross_probs=True)
    for b in range(2):
        logits, cross = om.decode_full(ids[b], enc[b: b + 1], want_attn=True)
        assert np.abs(out["logits_all"][b].cpu().numpy() - logits[0]).max() < tol_logits
        assert np.abs(out["logits_last"][b].cpu().numpy() - logits[0, -1]).max() < tol_logits
        got = out["cross_probs"][:, b].cpu().numpy()
        assert np.abs(got - np.stack([c[0] for c in cross])).max() < tol_probs
        assert np.allclose(got.sum(-1), 1.0, atol=1e-4)


@pytest.mark.parametrize("mode,tol", [("f32", 1e-4), ("bf16", 1e-2)])
def test_decode_step_continues_a_prefix(micro, mode, tol):
    """KV-cached single steps after a cache-less pass over the prefix reproduce the cache-less logits."""
    eng, om, enc = micro[mode], micro["ref"][mode], micro["enc"]
    eng.cross_kv(2, torch.from_numpy(enc).to(DEV))
    ids = [[824, 825, 826, 924, 5, 17, 900, 3, 321], [824, 826, 924, 9, 9, 9, 1000, 1, 2]]
    eng.decoder_forward(torch.tensor([r[:6] for r in ids], dtype=torch.int32), logits_last=True)
    for t in range(6, 9):
        got = eng.decode_step(torch.tensor([r[t] for r in ids]), torch.tensor([t, t])).cpu().numpy()
        for b in range(2):
            ref = om.decode_full(ids[b][: t + 1], enc[b: b + 1])[0, -1]
            assert np.abs(got[b] - ref).max() < tol, (t, b, np.abs(got[b] - ref).max())


def test_bf16_greedy_tokens_are_the_argmax_of_the_step_logits(micro):
    """The greedy loop's fused arg-max / bookkeeping kernel against the exported single step: every generated token must
    be exactly np.argmax (first index on ties) of the logits that uw_decode_step gives for the same prefix."""
    eng, enc = micro["bf16"], micro["enc"]
    sp = MICRO_SPECIALS
    eng.cross_kv(2, torch.from_numpy(enc).to(DEV))
    prompts = [[824, 825, 826, 924], [824, 826, 924, 9]]
    n_gen = 14
    res = eng.decode_greedy(prompts, len(prompts[0]) + n_gen, -1, sp["sot"], sp["ts_begin"], sp["ts_end"])   # eot never matches
    got = res["new_tokens"][:, :n_gen].cpu().numpy()
    assert (res["n_new"].cpu().numpy() == n_gen).all()
    eng.cross_kv(2, torch.from_numpy(enc).to(DEV))
    P = len(prompts[0])
    eng.decoder_forward(torch.tensor([r[: P - 1] for r in prompts], dtype=torch.int32), logits_last=True)
    last = [r[-1] for r in prompts]
    for t in range(n_gen):
        logits = eng.decode_step(torch.tensor(last), torch.tensor([P - 1 + t] * 2)).cpu().numpy()
        last = [int(np.argmax(logits[b])) for b in range(2)]
        assert last == got[:, t].tolist(), (t, last, got[:, t].tolist())


def _reference_window(om, enc, prompt, max_tokens, sp):
    new = O.greedy_reference_schedule(om, enc, prompt, max_tokens, sp["eot"])
    text = O.filter_text_tokens(new, sp["sot"], sp["ts_begin"], sp["ts_end"])
    rows = O.attention_rows_reference_schedule(om, enc, prompt, text) if text else None
    return new, text, rows


def test_greedy_tokens_rows_and_boundaries_match_reference_schedule(micro):
    """fp32 mode against the reference's schedule (cache-less re-decode per step, per-token attention replay):
    identical tokens, attention rows within 1e-6, DTW boundaries identical on the GPU's own cost matrix."""
    eng, om, enc = micro["f32"], micro["om"], micro["enc"]
    sp = dict(MICRO_SPECIALS, sot=700)      # many generated ids land above sot: exercises the teacher-forced fix-up
    eng.cross_kv(2, torch.from_numpy(enc).to(DEV))
    prompts = [[824, 825, 826, 924], [823, 5, 6, 7, 824, 825, 826, 924, 11, 12]]     # ragged lengths
    max_tokens = 40
    res = eng.decode_greedy(prompts, max_tokens, sp["eot"], sp["sot"], sp["ts_begin"], sp["ts_end"])
    bnd = eng.align(res["n_text"]).cpu().numpy()
    saw_fixup = False
    for b, pr in enumerate(prompts):
        new, text, rows_ref = _reference_window(om, enc[b: b + 1], pr, max_tokens, sp)
        nn, nt = int(res["n_new"][b]), int(res["n_text"][b])
        assert res["new_tokens"][b, :nn].cpu().tolist() == new
        assert res["text_ids"][b, :nt].cpu().tolist() == text
        assert len(new) == max_tokens - len(pr)
        first_special = next((i for i, t in enumerate(new) if t not in text), None)
        if first_special is not None and any(t in text for t in new[first_special:]):
            saw_fixup = True
        if text:
            rows = eng.attn_rows(2, nt)[b].cpu().numpy()
            ref_sel = np.stack([rows_ref[l][0, h] for l, h in HEADS])
            assert np.abs(rows - ref_sel).max() < 1e-6
            cost = E.cost_matrix(torch.from_numpy(rows[None]).to(DEV), torch.tensor([nt], dtype=torch.int32)).cpu().numpy()[0]
            i1, i2 = O.dtw_symmetric1(cost)
            assert np.array_equal(bnd[b, : nt + 1], O.token_boundaries(i1, i2, nt))
            assert np.allclose(cost, O.cost_matrix(rows_ref, HEADS), rtol=5e-5, atol=0)
    assert saw_fixup, "test inputs no longer produce text after a non-text token"


def test_greedy_stops_on_eot_and_budget(micro):
    eng, om, enc = micro["f32"], micro["om"], micro["enc"]
    eng.cross_kv(2, torch.from_numpy(enc).to(DEV))
    prompt = [824, 825, 826, 924]
    free = O.greedy_reference_schedule(om, enc[:1], prompt, 30, eot=-1)
    eot = free[6]                                         # declare the 7th generated id to be end-of-text
    k = free.index(eot)
    res = eng.decode_greedy([prompt, prompt], 30, eot, 10 ** 6, 10 ** 6, 10 ** 6)
    assert int(res["n_new"][0]) == k and res["new_tokens"][0, :k].cpu().tolist() == free[:k]
    res = eng.decode_greedy([prompt], len(prompt), -1, 10 ** 6, 10 ** 6, 10 ** 6)      # no budget: nothing generated
    assert int(res["n_new"][0]) == 0 and int(res["n_text"][0]) == 0
    res = eng.decode_greedy([prompt], len(prompt) + 3, -1, 10 ** 6, 10 ** 6, 10 ** 6)
    assert res["new_tokens"][0, :3].cpu().tolist() == free[:3] and int(res["n_new"][0]) == 3


def test_bf16_teacher_forced_agreement(micro):
    """bf16 is compared on the fp32 token stream: logits tolerance and argmax agreement rate."""
    eng, om, enc = micro["bf16"], micro["om"], micro["enc"]
    eng.cross_kv(1, torch.from_numpy(enc[:1]).to(DEV))
    prompt = [824, 825, 826, 924]
    seq = prompt + O.greedy_reference_schedule(om, enc[:1], prompt, 36, eot=-1)
    ref32 = om.decode_full(seq, enc[:1])[0]
    ref = micro["ref"]["bf16"].decode_full(seq, enc[:1])[0]
    got = eng.decoder_forward(torch.tensor([seq], dtype=torch.int32), logits_all=True)["logits_all"][0].cpu().numpy()
    assert np.abs(got - ref).max() < 1e-2, np.abs(got - ref).max()
    assert np.abs(got - ref32).max() < np.abs(ref - ref32).max() + 1e-2
    agree = (got.argmax(-1) == ref32.argmax(-1)).mean()
    assert agree > 0.9, agree


def test_transcribe_windows_end_to_end(micro):
    """audio -> tokens + boundary frames in one call equals the staged calls and the oracle's tokens."""
    eng, om = micro["f32"], micro["om"]
    sp = MICRO_SPECIALS
    a0, a1 = micro["audio"]
    buf = np.zeros((2, 480000), np.float32)
    buf[0] = a0
    buf[1, : len(a1)] = a1
    prompts = [[824, 825, 826, 924], [824, 825, 826, 924, 3]]
    res = eng.transcribe_windows(torch.from_numpy(buf).to(DEV), torch.tensor([480000, len(a1)], dtype=torch.int32), prompts,
                                 24, sp["eot"], sp["sot"], sp["ts_begin"], sp["ts_end"])
    for b, pr in enumerate(prompts):
        new = O.greedy_reference_schedule(om, micro["enc"][b: b + 1], pr, 24, sp["eot"])
        nn, nt = int(res["n_new"][b]), int(res["n_text"][b])
        assert res["new_tokens"][b, :nn].cpu().tolist() == new
        bd = res["boundaries"][b, : nt + 1].cpu().numpy()
        assert bd[0] == 0 and np.all(np.diff(bd) >= 0) and bd[-1] <= 1500
        assert bd[-1] == bd[-2]        # the reference's length fix-up repeats the last boundary


def test_batches_beyond_128_windows_decode_in_chunks():
    """The decode step's products take at most 128 rows per launch; 160 windows must give what 2 windows give."""
    dims = CFG.dims_for("micro")
    sd = synthetic_state_dict(dims, seed=0, std=0.06)
    B = 160
    eng = E.Engine(dims, sd, dtype="f32", max_batch=B, align_heads=HEADS)
    try:
        base = [synth_audio(1), synth_audio(2, 300000)]
        buf = np.zeros((B, 480000), np.float32)
        for b in range(B):
            buf[b, : len(base[b % 2])] = base[b % 2]
        n = torch.tensor([len(base[b % 2]) for b in range(B)], dtype=torch.int32)
        sp = MICRO_SPECIALS
        res = eng.transcribe_windows(torch.from_numpy(buf).to(DEV), n, [[824, 825, 826, 924]] * B, 16, sp["eot"], sp["sot"],
                                     sp["ts_begin"], sp["ts_end"])
        toks, bnd = res["new_tokens"].cpu().numpy(), res["boundaries"].cpu().numpy()
        assert (res["n_new"].cpu().numpy() == 12).all()
        for b in range(2, B):
            assert np.array_equal(toks[b], toks[b % 2]) and np.array_equal(bnd[b], bnd[b % 2]), b
    finally:
        eng.close()


def test_cross_attention_decode_forms_agree(micro):
    """The one-workgroup-per-(window, head) form (what full batches run) and the row-split form (small batches) of the
    cross-attention decode step: same tokens, rows within 1e-6 of each other and of the oracle."""
    import os
    om, enc = micro["om"], micro["enc"]
    sp = dict(MICRO_SPECIALS, sot=700)
    prompts = [[824, 825, 826, 924], [823, 5, 6, 7, 824, 825, 826, 924, 11, 12]]
    got = {}
    for mode in ("0", "3", "8"):
        os.environ["UW_DECODE_SPLIT"] = mode
        try:
            eng = E.Engine(micro["dims"], micro["sd"], dtype="f32", max_batch=2, align_heads=HEADS)
        finally:
            del os.environ["UW_DECODE_SPLIT"]
        try:
            eng.cross_kv(2, torch.from_numpy(enc).to(DEV))
            res = eng.decode_greedy(prompts, 30, sp["eot"], sp["sot"], sp["ts_begin"], sp["ts_end"])
            nt = res["n_text"].cpu().numpy()
            got[mode] = (res["new_tokens"].cpu().numpy(), nt, eng.attn_rows(2, int(nt.max())).cpu().numpy())
        finally:
            eng.close()
    new, text, rows_ref = _reference_window(om, enc[:1], prompts[0], 30, sp)
    for mode in ("0", "3", "8"):
        toks, nt, rows = got[mode]
        assert np.array_equal(toks, got["0"][0]) and np.array_equal(nt, got["0"][1])
        assert toks[0, : len(new)].tolist() == new
        assert np.abs(rows - got["0"][2]).max() < 1e-6
        assert np.abs(rows[0, :, : len(text)] - np.stack([rows_ref[l][0, h] for l, h in HEADS])).max() < 1e-6


def test_weight_prefetch_changes_no_result(micro):
    """Decode-step products may carry extra workgroups that request the following product's weights (used up to 32
    windows by default).  They compute nothing: tokens, logits-derived rows and counters are bit-identical with the
    prefetch forced on and off, for one row of workgroups (2 windows) and several (40 windows, padded grids), in both
    compute types."""
    import os
    enc = micro["enc"]
    sp = dict(MICRO_SPECIALS, sot=700)
    base = [[824, 825, 826, 924], [823, 5, 6, 7, 824, 825, 826, 924, 11, 12]]
    for dtype in ("f32", "bf16"):
        for B in (2, 40):
            prompts = [base[i % 2] for i in range(B)]
            enc_b = torch.from_numpy(np.concatenate([enc[i % 2: i % 2 + 1] for i in range(B)])).to(DEV)
            got = {}
            for mode in ("0", "1"):
                os.environ["UW_STEP_PREFETCH"] = mode
                try:
                    eng = E.Engine(micro["dims"], micro["sd"], dtype=dtype, max_batch=B, align_heads=HEADS)
                finally:
                    del os.environ["UW_STEP_PREFETCH"]
                try:
                    eng.cross_kv(B, enc_b)
                    res = eng.decode_greedy(prompts, 30, sp["eot"], sp["sot"], sp["ts_begin"], sp["ts_end"])
                    nt = res["n_text"].cpu().numpy()
                    got[mode] = (res["new_tokens"].cpu().numpy(), nt, eng.attn_rows(B, int(nt.max())).cpu().numpy())
                finally:
                    eng.close()
            for a, b in zip(got["0"], got["1"]):
                assert np.array_equal(a, b), (dtype, B)
            assert got["1"][1].max() > 0


def test_one_launch_logits_agree_with_separate_kernels(micro):
    """Up to 16 windows in bf16, a decode step produces its logits in one launch (final LayerNorm, tied projection and
    per-block arg-max candidates, uw_logits_rows16) instead of LayerNorm + projection + a scan of the row.  Both forms:
    step logits within 2e-3 of each other (the LayerNorm is evaluated in a different order), the arg-max of every row
    found through the candidates equals the arg-max of the stored logits, and greedy decoding gives the same tokens."""
    import os
    enc = micro["enc"]
    sp = dict(MICRO_SPECIALS, sot=700)
    base = [[824, 825, 826, 924], [823, 5, 6, 7, 824, 825, 826, 924, 11, 12]]
    for B in (1, 3, 16):
        prompts = [base[i % 2] for i in range(B)]
        enc_b = torch.from_numpy(np.concatenate([enc[i % 2: i % 2 + 1] for i in range(B)])).to(DEV)
        got = {}
        for mode in ("0", "1"):
            os.environ["UW_LOGITS_FUSED"] = mode
            try:
                eng = E.Engine(micro["dims"], micro["sd"], dtype="bf16", max_batch=B, align_heads=HEADS)
            finally:
                del os.environ["UW_LOGITS_FUSED"]
            try:
                eng.cross_kv(B, enc_b)
                ids = torch.tensor([p[0] for p in prompts], dtype=torch.int32)
                logits = eng.decode_step(ids, torch.zeros(B, dtype=torch.int32)).cpu().numpy()
                res = eng.decode_greedy(prompts, 30, sp["eot"], sp["sot"], sp["ts_begin"], sp["ts_end"])
                got[mode] = (logits, res["new_tokens"].cpu().numpy(), res["n_new"].cpu().numpy())
            finally:
                eng.close()
        assert np.isfinite(got["1"][0]).all()
        assert np.abs(got["0"][0] - got["1"][0]).max() < 2e-3, (B, np.abs(got["0"][0] - got["1"][0]).max())
        assert np.array_equal(got["0"][1], got["1"][1]) and np.array_equal(got["0"][2], got["1"][2]), B
        assert got["1"][2].max() > 0


def test_row_split_merge_is_run_to_run_identical(micro):
    """The row-split cross-attention merges its parts in whichever workgroup finishes last (tickets, device-scope loads
    and stores).  The merge order is fixed, so repeated runs must give bit-identical tokens and attention rows, in both
    compute types, whichever workgroup that was."""
    import os
    enc = micro["enc"]
    sp = dict(MICRO_SPECIALS, sot=700)
    prompts = [[824, 825, 826, 924], [823, 5, 6, 7, 824, 825, 826, 924, 11, 12]]
    for dtype in ("f32", "bf16"):
        os.environ["UW_DECODE_SPLIT"] = "8"
        try:
            eng = E.Engine(micro["dims"], micro["sd"], dtype=dtype, max_batch=2, align_heads=HEADS)
        finally:
            del os.environ["UW_DECODE_SPLIT"]
        try:
            eng.cross_kv(2, torch.from_numpy(enc).to(DEV))
            first = None
            for _ in range(40):
                res = eng.decode_greedy(prompts, 30, sp["eot"], sp["sot"], sp["ts_begin"], sp["ts_end"])
                nt = res["n_text"].cpu().numpy()
                got = (res["new_tokens"].cpu().numpy(), nt, eng.attn_rows(2, int(nt.max())).cpu().numpy())
                if first is None:
                    first = got
                    assert nt.max() > 0 and np.isfinite(got[2]).all()
                else:
                    for a, b in zip(first, got):
                        assert np.array_equal(a, b), dtype
        finally:
            eng.close()
