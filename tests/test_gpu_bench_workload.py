"""The benchmark's own workload against the oracle: whisper-small shapes, 128 windows, the bench's 170-token
diarization-style prompt, N(0, 0.02^2) weights.

  * bf16 (what the bench times), teacher-forced on the fp32 oracle's greedy token stream: logits of windows 0, 63 and
    127 from the multi-token pass over prompt + 16 tokens (the 23,808-row prefill products, causal attention over 186
    positions) and from 16 KV-cached steps at self-attention lengths 171..186 after a 170-token prefill (the step
    kernels at 128 rows).  Tolerance: 1e-2 abs, the contract's bf16 figure, against the oracle evaluated in fp32 on
    the weights the bf16 engine holds (tests.helpers.bf16_weights).  Against the oracle on the ORIGINAL fp32 weights
    the distance is bounded by fixed figures (what was measured, plus 10 %): at this depth rounding the weights alone
    moves the fp32 oracle's logits by about 1.5e-2, so no bf16-weight implementation can be within 1e-2 of the
    fp32-weight logits.
  * bf16 free-running (the run the bench times) against fp32 free-running: greedy tokens of windows 0, 63 and 127 over
    128 generated tokens -- the agreement rate and the first diverging step are reported; with random weights the two
    best logits are often closer than the bf16 error, and one flip changes everything after it.
  * fp32: greedy tokens from the 170-token prompt identical to the reference's cache-less schedule on two different
    windows, attention rows within 1e-5, boundaries identical to the oracle's DTW on the GPU's cost matrix.
"""
import numpy as np
import pytest
import torch

from bench import EOT, SOT, TS_BEGIN, TS_END, make_prompt
from oracle import whisper_oracle as O
from tests.helpers import bf16_weights, synth_audio
from whisper_amd import engine as E
from whisper_amd.config import alignment_heads, alignment_heads_key, dims_for
from whisper_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF16_LOGITS_TOL = 1e-2
FULL32_BOUND, STEP32_BOUND = 1.76e-2, 1.57e-2     # bf16 engine vs the oracle on fp32 weights: measured 1.60e-2 / 1.42e-2, + 10 %


@pytest.fixture(scope="module")
def small():
    dims = dims_for("small")
    sd = synthetic_state_dict(dims, seed=0)
    heads = alignment_heads(alignment_heads_key("small", dims.d_model, "en"), dims.dec_layers, dims.n_heads)
    return dims, sd, heads, O.OracleModel(sd, dims.n_heads)


def test_bench_workload_bf16_logits_within_contract(small):
    dims, sd, heads, om = small
    om_r = O.OracleModel(bf16_weights(sd), dims.n_heads)
    B, n_tf = 128, 16
    prompt = make_prompt(0)
    P = len(prompt)
    assert P == 170
    check = (0, 63, 127)
    audio = np.stack([synth_audio(1234 + b) for b in range(B)])
    seqs, refs, refs32 = {}, {}, {}
    floor = 0.0
    for b in check:
        mel_ref = O.log_mel_segment(audio[b], dims.n_mels)
        enc_ref = om.encode(mel_ref)
        new = O.greedy_reference_schedule(om, enc_ref, prompt, P + n_tf, EOT)
        assert len(new) == n_tf
        seqs[b] = prompt + new
        refs32[b] = om.decode_full(seqs[b], enc_ref)[0]                # [P + n_tf, V], fp32 weights
        refs[b] = om_r.decode_full(seqs[b], om_r.encode(mel_ref))[0]   # fp32 arithmetic on the engine's bf16 weights
        floor = max(floor, float(np.abs(refs[b] - refs32[b]).max()))
    ids = np.array([seqs[b] if b in seqs else seqs[0] for b in range(B)], dtype=np.int32)
    eng = E.Engine(dims, sd, dtype="bf16", max_batch=B, align_heads=heads)
    try:
        mel = eng.mel(torch.from_numpy(audio).to(DEV), torch.full((B,), 480000, dtype=torch.int32))
        eng.encode(mel)
        eng.cross_kv(B)
        full = eng.decoder_forward(torch.from_numpy(ids), logits_all=True, logits_last=False)["logits_all"]
        worst_full = max(float(np.abs(full[b].cpu().numpy() - refs[b]).max()) for b in check)
        worst_full32 = max(float(np.abs(full[b].cpu().numpy() - refs32[b]).max()) for b in check)
        del full
        # the bench's schedule: prompt pass, then KV-cached steps with 128 rows per product
        last = eng.decoder_forward(torch.from_numpy(ids[:, :P].copy()), logits_last=True)["logits_last"].cpu().numpy()
        worst_step = max(float(np.abs(last[b] - refs[b][P - 1]).max()) for b in check)       # the prompt pass's own logits
        worst_step32 = max(float(np.abs(last[b] - refs32[b][P - 1]).max()) for b in check)
        agree = []
        for t in range(P, P + n_tf):
            got = eng.decode_step(torch.from_numpy(ids[:, t].copy()), torch.full((B,), t, dtype=torch.int32)).cpu().numpy()
            for b in check:
                worst_step = max(worst_step, float(np.abs(got[b] - refs[b][t]).max()))
                worst_step32 = max(worst_step32, float(np.abs(got[b] - refs32[b][t]).max()))
                agree.append(int(got[b].argmax()) == int(refs32[b][t].argmax()))
        print(f"small B=128 P=170 bf16 max|dlogit|: multi-token pass {worst_full:.2e} / cached steps {worst_step:.2e} vs the fp32 oracle on "
              f"the engine's bf16 weights; {worst_full32:.2e} / {worst_step32:.2e} vs the oracle on fp32 weights (weight rounding alone "
              f"moves the oracle by {floor:.2e}); argmax agreement with the fp32 oracle {np.mean(agree):.2f}")
        assert worst_full <= BF16_LOGITS_TOL, worst_full
        assert worst_step <= BF16_LOGITS_TOL, worst_step
        # distance to the fp32-WEIGHT oracle, the literal reading of the contract: above 1e-2 (rounding the weights alone
        # moves the oracle by 1.49e-2 here), pinned at what was measured on MI355X (1.60e-2 / 1.42e-2) plus 10 % so that it
        # cannot drift upwards unnoticed
        assert abs(floor - 1.49e-2) < 1e-3, floor
        assert worst_full32 <= FULL32_BOUND and worst_step32 <= STEP32_BOUND, (worst_full32, worst_step32)
    finally:
        eng.close()


def test_bench_prompt_fp32_tokens_rows_boundaries(small):
    dims, sd, heads, om = small
    B, n_gen = 48, 8           # 48 x 12 heads = 576 (window, head) pairs: the decode-attention form full batches run
    prompt = make_prompt(0)
    P = len(prompt)
    a0, a1 = synth_audio(1234), synth_audio(1234 + 127)
    audio = np.repeat(a0[None], B, 0)
    audio[B - 1] = a1
    eng = E.Engine(dims, sd, dtype="f32", max_batch=B, align_heads=heads)
    try:
        res = eng.transcribe_windows(torch.from_numpy(audio).to(DEV), torch.full((B,), 480000, dtype=torch.int32), [prompt] * B,
                                     P + n_gen, EOT, SOT, TS_BEGIN, TS_END)
        bnd = res["boundaries"].cpu().numpy()
        for b, a in ((0, a0), (B - 1, a1)):
            enc_ref = om.encode(O.log_mel_segment(a, dims.n_mels))
            new = O.greedy_reference_schedule(om, enc_ref, prompt, P + n_gen, EOT)
            text = O.filter_text_tokens(new, SOT, TS_BEGIN, TS_END)
            nn, nt = int(res["n_new"][b]), int(res["n_text"][b])
            assert res["new_tokens"][b, :nn].cpu().tolist() == new, b
            assert res["text_ids"][b, :nt].cpu().tolist() == text
            if text:
                rows = eng.attn_rows(B, nt)[b].cpu().numpy()
                rows_ref = O.attention_rows_teacher_forced(om, enc_ref, prompt, text)
                assert np.abs(rows - np.stack([rows_ref[l][0, h] for l, h in heads])).max() < 1e-5
                cost = E.cost_matrix(torch.from_numpy(rows[None]).to(DEV), torch.tensor([nt], dtype=torch.int32)).cpu().numpy()[0]
                i1, i2 = O.dtw_symmetric1(cost)
                assert np.array_equal(bnd[b, : nt + 1], O.token_boundaries(i1, i2, nt))
    finally:
        eng.close()


def test_bench_workload_free_running_token_agreement(small):
    """The bench's own run (bf16, 128 windows, 170-token prompt, 128 generated tokens, free-running) beside the same
    windows decoded in fp32 (whose tokens equal the oracle's, see the test above): how far do the token streams agree?"""
    dims, sd, heads, om = small
    B, n_gen = 128, 128
    prompt = make_prompt(0)
    P = len(prompt)
    check = (0, 63, 127)
    audio = np.stack([synth_audio(1234 + b) for b in range(B)])
    n = torch.full((B,), 480000, dtype=torch.int32)
    eng = E.Engine(dims, sd, dtype="bf16", max_batch=B, align_heads=heads)
    try:
        res = eng.transcribe_windows(torch.from_numpy(audio).to(DEV), n, [prompt] * B, P + n_gen, EOT, SOT, TS_BEGIN, TS_END)
        got = {b: res["new_tokens"][b, : int(res["n_new"][b])].cpu().tolist() for b in check}
    finally:
        eng.close()
    eng = E.Engine(dims, sd, dtype="f32", max_batch=len(check), align_heads=heads)
    try:
        res = eng.transcribe_windows(torch.from_numpy(audio[list(check)]).to(DEV), n[: len(check)], [prompt] * len(check), P + n_gen,
                                     EOT, SOT, TS_BEGIN, TS_END)
        ref = {b: res["new_tokens"][i, : int(res["n_new"][i])].cpu().tolist() for i, b in enumerate(check)}
    finally:
        eng.close()
    first, rate = {}, {}
    for b in check:
        m = min(len(got[b]), len(ref[b]))
        same = [got[b][i] == ref[b][i] for i in range(m)]
        first[b] = same.index(False) if False in same else m
        rate[b] = float(np.mean(same)) if m else 1.0
    print(f"small B=128 P=170 free-running greedy, bf16 vs fp32: tokens equal up to step {first} of {n_gen}; agreement rate {rate}")
    # every window starts out on the fp32 stream; how long it stays there depends on the margins of random weights
    assert all(len(got[b]) == n_gen for b in check)
    assert min(first.values()) >= 1, first
