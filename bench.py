#!/usr/bin/env python3
"""Benchmark of the transcription hot path on MI355X.

One "step" = one pass of the whole path (log-mel -> encoder -> cross K/V -> KV-cached greedy decode with
attention-row capture -> teacher-forced row fix-up -> cost matrix -> DTW) over one batch of synthetic 30 s
windows per GPU, inputs resident in HBM.  Metric: audio-seconds per second (RTFx), whole job.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong] [--inflight I]

Every GPU keeps --inflight batches in flight (whisper_amd.parallel.EnginePool: a context, a HIP stream and a host thread
per batch), because the kernels of independent batches fill each other's gaps; the K timed steps are K batches pushed
through that pool.  Per-kernel durations only mean something when the kernel has the GPU to itself, so the `roofline`
figures come from another timed region of the same run (ahead of the headline one) in which the same batches go through
one at a time; the figures of the headline region are reported beside them under `overlapped`.

With --gpus N > 1 and no torch.distributed environment the script starts its own N rank processes (one GPU each,
before the parent touches any GPU); under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
it uses the ranks it is given.  Independent windows shard with no data-path collective; the only exchange is the
final all_gather of token / boundary tensors (RCCL).
  weak   (default): every rank transcribes its own --batch windows.
  strong          : --global-batch windows are dealt out over the ranks (whisper_amd.parallel.shard_windows).
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SOT, EOT, LANG_EN, TRANSCRIBE, TS_BEGIN, TS_END, SOT_PREV = 50258, 50257, 50259, 50359, 50364, 51864, 50361
HBM_PEAK_GBS, HBM_STREAM_GBS, MFMA_BF16_PEAK_TF = 8000.0, 6300.0, 2500.0


def make_prompt(window_idx: int, n_prompt_ids: int = 160, n_prefix: int = 5):
    """[sot_prev] + fixed pseudo prompt ids + [sot, lang, transcribe, <|0.00|>] + prefix ids: stands in for the
    tokenised SPEAKERS(...) prompt and a ' [SPEAKER 1]' prefix (core.py:874-922).  170 tokens."""
    rng = np.random.default_rng(99)          # the same prompt text for every window
    body = rng.integers(1000, 40000, size=n_prompt_ids).tolist()
    prefix = rng.integers(1000, 40000, size=n_prefix).tolist()
    return [SOT_PREV] + body + [SOT, LANG_EN, TRANSCRIBE, TS_BEGIN] + prefix


def synth_audio_gpu(seeds, device):
    """0.1 N(0,1) noise + a 200 -> 3000 Hz chirp per window, generated on the device, one generator seed per window."""
    import torch
    t = torch.arange(480000, device=device, dtype=torch.float32) / 16000.0
    chirp = 0.3 * torch.sin(2 * np.pi * (200 * t + (2800 / 60.0) * t * t))
    out = torch.empty((len(seeds), 480000), device=device, dtype=torch.float32)
    g = torch.Generator(device=device)
    for i, seed in enumerate(seeds):
        g.manual_seed(int(seed))
        out[i] = 0.1 * torch.randn(480000, generator=g, device=device, dtype=torch.float32) + chirp
    return out.contiguous()


def path_roofline_ms(dims, B, P, G, es):
    """Lower bound of one pass from the path's algorithmic work: dense contractions (encoder, cross K/V projection,
    prompt pass) at the bf16 MFMA peak, the decode steps at the achievable HBM streaming rate (weights once per step,
    cross K/V and the self K/V cache once per window and step)."""
    d, Le, Ld, V, nm = dims.d_model, dims.enc_layers, dims.dec_layers, dims.vocab, dims.n_mels
    enc = 2 * 3000 * d * 3 * nm + 2 * 1500 * d * 3 * d + Le * (36000 * d * d + 9e6 * d)
    cross_proj = Ld * 6000 * d * d
    tok = lambda t: Ld * (28 * d * d + 4 * t * d + 6000 * d) + 2 * d * V          # decoder flops of one token at length t
    prefill = P * tok(P / 2)
    dense_s = B * (enc + cross_proj + prefill) / (MFMA_BF16_PEAK_TF * 1e12)
    w_dec = (Ld * 14 * d * d + V * d) * es
    kv_cross = Ld * 2 * 1500 * d * es
    step_bytes = sum(w_dec + B * (kv_cross + Ld * 2 * d * es * (P + k)) for k in range(G))
    return 1e3 * (dense_s + step_bytes / (HBM_STREAM_GBS * 1e9))


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(dims, sd, heads, prompt, n_gen: int, n_windows: int = 2, n_sample_steps: int = 24):
    """The oracle (CPU restatement of the reference) on windows of the same workload, bounded.
    Reference schedule (core.py:938-968, align.py:607-623): per window log-mel + encoder once, `n_sample_steps` of the
    n_gen cache-less greedy passes and as many of the n_gen attention-replay passes (prefix lengths spread evenly over
    the real range) scaled linearly, cost matrix and DTW; `value` is the median window.  `opt` repeats one window with
    the schedule an optimised CPU implementation would use (K/V cached, rows from the decode steps)."""
    import torch
    from oracle import whisper_oracle as O
    om = O.OracleModel(sd, dims.n_heads)
    P = len(prompt)
    lens = np.linspace(P, P + n_gen - 1, n_sample_steps).round().astype(int)
    scale = n_gen / n_sample_steps
    dtw_c = None
    try:
        import ctypes
        dtw_c = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libdtw_ref.so"))
    except OSError:
        pass

    def align_seconds(att):
        t0 = time.perf_counter()
        cost = O.cost_matrix(att, heads)
        if dtw_c is not None:
            n, m = cost.shape
            pi = np.zeros(n + m, np.int32); pj = np.zeros(n + m, np.int32); bd = np.zeros(n + 1, np.int32)
            dtw_c.dtw_symmetric1_ref(cost.ctypes.data_as(ctypes.c_void_p), n, m, pi.ctypes.data_as(ctypes.c_void_p),
                                     pj.ctypes.data_as(ctypes.c_void_p), bd.ctypes.data_as(ctypes.c_void_p), 1500)
        return time.perf_counter() - t0

    per_window, parts = [], []
    enc = None
    for w in range(n_windows):
        rng = np.random.default_rng(1234 + w)
        audio = (0.1 * rng.standard_normal(480000)).astype(np.float32)
        t0 = time.perf_counter()
        enc = om.encode(O.log_mel_segment(audio, dims.n_mels))
        t_front = time.perf_counter() - t0
        filler = rng.integers(1000, 40000, size=P + n_gen).tolist()
        t_logits = t_attn = 0.0
        for T in lens:
            ids = (list(prompt) + filler)[:T]
            t0 = time.perf_counter()
            int(np.argmax(om.decode_full(ids, enc)[0, -1]))
            t_logits += time.perf_counter() - t0
            t0 = time.perf_counter()
            om.decode_full(ids, enc, want_attn=True)
            t_attn += time.perf_counter() - t0
        att = [np.random.default_rng(5).random((1, dims.n_heads, n_gen, 1500), dtype=np.float32) for _ in range(dims.dec_layers)]
        t_align = align_seconds(att)
        per_window.append(t_front + scale * (t_logits + t_attn) + t_align)
        parts.append((t_front, t_logits, t_attn, t_align))
    ref_s = float(np.median(per_window))
    # better schedule, one window: encoder as timed above, K/V-cached prompt pass + every fourth of the n_gen steps
    t_front = parts[-1][0]
    dec = O.CachedDecoder(om, enc)
    t0 = time.perf_counter()
    dec.forward(prompt)
    t_prefill = time.perf_counter() - t0
    n_opt = max(1, n_gen // 4)
    t0 = time.perf_counter()
    for k in range(n_opt):
        lg, _ = dec.forward([1000 + k])
        int(np.argmax(lg[0]))
    t_steps = (time.perf_counter() - t0) * (n_gen / n_opt)
    opt_s = t_front + t_prefill + t_steps + parts[-1][3]
    threads = int(torch.get_num_threads())
    f, lg, at, al = parts[-1]
    return {"value": 30.0 / ref_s, "unit": "audio-sec/sec", "cores": threads, "kind": "port",
            "cpu": cpu_model_name(), "host_cpus": os.cpu_count(),
            "opt": {"value": 30.0 / opt_s, "unit": "audio-sec/sec",
                    "sample": (f"same oracle with the K/V-cached schedule on 1 window: encoder {t_front:.2f}s, prompt pass {t_prefill:.2f}s, "
                               f"{n_opt} of {n_gen} cached steps scaled x{n_gen / n_opt:.0f} ({t_steps:.2f}s), alignment {al:.3f}s")},
            "sample": (f"median of {n_windows} windows of the same config on the host CPU with the reference's cache-less schedule "
                       f"(the reference's onnxruntime path cannot run here: onnxruntime and ONNX files are absent, so the CPU restatement "
                       f"is timed): per window log-mel + encoder once ({f:.2f}s), {n_sample_steps} of {n_gen} greedy decoder passes ({lg:.2f}s) "
                       f"and {n_sample_steps} of {n_gen} attention-replay passes ({at:.2f}s) at prefix lengths {int(lens[0])}..{int(lens[-1])}, "
                       f"scaled x{scale:.1f}; cost matrix + DTW {al:.3f}s; torch fp32, {threads} threads; per-window seconds "
                       f"{[round(x, 1) for x in per_window]}")}


def spawn_ranks(n_ranks, argv, script=None, poll_s=0.2, grace_s=5.0):
    """--gpus N without a torch.distributed environment: start N rank processes, one GPU each.  The parent has not
    touched a GPU (no HIP call, no torch.cuda query); it polls all children, and when one exits non-zero it stops the
    others (they would otherwise sit in the rendezvous or a barrier until the caller's time limit) and returns that
    rank's exit code.  Rank 0's output passes through."""
    port = 29500 + os.getpid() % 2000
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env))
    failed = 0
    live = set(range(n_ranks))
    while live and not failed:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                failed = abs(rc) or 1
                print(f"bench: rank {r} exited with code {rc}; stopping the other ranks", file=sys.stderr, flush=True)
                break
        if live and not failed:
            time.sleep(poll_s)
    if failed:
        for r in live:
            procs[r].terminate()                 # exact child processes this call started
        deadline = time.time() + grace_s
        for r in live:
            try:
                procs[r].wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
    return failed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="small")
    ap.add_argument("--batch", type=int, default=128, help="windows per GPU (weak scaling)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--global-batch", type=int, default=128, help="windows of the whole job (strong scaling)")
    ap.add_argument("--gen", type=int, default=128, help="tokens generated per window")
    ap.add_argument("--inflight", type=int, default=3, help="batches in flight per GPU (1 = one batch at a time)")
    ap.add_argument("--solo-steps", type=int, default=3, help="steps of the one-at-a-time region behind the roofline figures")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo only to rehearse N > 1 on one GPU)")
    ap.add_argument("--device", type=int, default=None, help="GPU index for every rank (rehearsal only; default LOCAL_RANK)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.device is not None:
        local = args.device

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    from whisper_amd import engine as E
    from whisper_amd.config import alignment_heads, alignment_heads_key, dims_for
    from whisper_amd.parallel import EnginePool, ResultExchange, shard_windows
    from whisper_amd.weights import synthetic_state_dict

    dims = dims_for(args.model)
    sd = synthetic_state_dict(dims, seed=0)
    heads = alignment_heads(alignment_heads_key(args.model, dims.d_model, "en"), dims.dec_layers, dims.n_heads)
    if args.scaling == "strong":
        n_global = args.global_batch
        mine = shard_windows(n_global, rank, world)
    else:
        n_global = args.batch * world
        mine = range(rank * args.batch, (rank + 1) * args.batch)
    B = len(mine)
    if B == 0:
        raise SystemExit(f"rank {rank} has no windows: --global-batch {n_global} < {world} ranks")
    kw = dict(dtype=args.dtype, max_batch=B, align_heads=heads, device=f"cuda:{local}")
    pool = EnginePool(dims, sd, inflight=args.inflight, **kw) if args.inflight > 1 else None
    eng = pool.engines[0] if pool else E.Engine(dims, sd, **kw)     # one batch at a time: no pool, no worker thread
    prompt = make_prompt(0)
    P = len(prompt)
    max_tokens = P + args.gen
    prompts = [prompt] * B
    audio = synth_audio_gpu([1234 + w for w in mine], device)
    n_samples = torch.full((B,), 480000, dtype=torch.int32, device=device)
    call = (audio, n_samples, prompts, max_tokens, EOT, SOT, TS_BEGIN, TS_END)

    exchange = ResultExchange(n_global, rank, world)   # preallocated buffers, one all_gather per step

    def finish(res):
        if world > 1:   # the path's only exchange: gather tokens and boundary frames on every rank
            exchange.gather({k: res[k] for k in ("new_tokens", "boundaries", "n_new", "n_text")})
        return res

    def run_steps(k):
        """k batches through the pool, `inflight` at a time; results are gathered in step order."""
        res = None
        if pool is None:
            for _ in range(k):
                res = finish(eng.transcribe_windows(*call))
            return res
        futs = [pool.transcribe_windows_async(*call) for _ in range(k)]
        for f in futs:
            res = finish(f.result())
        return res

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(fn):
        fence()
        t0 = time.perf_counter()
        res = fn()
        fence()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return res, float(t.item())

    for e in (pool.engines if pool else [eng]):   # every context warms up (first-launch costs, allocator)
        for _ in range(max(1, args.warmup)):
            finish(e.transcribe_windows(*call))
    # ---- region 1: one batch at a time on one context -> per-kernel durations (roofline) ----
    solo_steps = max(1, args.solo_steps)
    eng.profile_begin(16384)
    _, solo_elapsed = timed(lambda: [finish(eng.transcribe_windows(*call)) for _ in range(solo_steps)][-1])
    prof = eng.profile_end()
    # ---- region 2, the headline: exactly --steps batches through the pool ----
    eng.profile_begin(16384)
    res, elapsed = timed(lambda: run_steps(args.steps))
    prof_ov = eng.profile_end()

    n_new = res["n_new"].cpu().numpy()
    n_text = res["n_text"].cpu().numpy()
    value = 30.0 * n_global * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    es = 2 if args.dtype == "bf16" else 4
    d = dims.d_model
    # algorithmic bytes of ONE cross-attention decode launch: K and V of one decoder layer for the windows it covers
    windows_per_launch = prof["windows"] / max(1, prof["launches"])
    bytes_per_launch = windows_per_launch * 2 * 1500 * d * es
    avg_ms = prof["timed_ms"] / max(1, prof["timed"])
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM bytes per launch of the same kernel and the encoder's MFMA utilisation from the PMC counters (separate
    # rocprofv3 --pmc passes, summary committed under profiles/); quoted only when measured on this very configuration
    traffic, mfma_busy = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_cross_decode.json")), reverse=True):
        try:
            with open(path) as fh:
                pmc = json.load(fh)
            if (pmc["model"], pmc["batch"], pmc["dtype"]) == (args.model, B, args.dtype) and abs(windows_per_launch - B) < 1e-9:
                traffic = pmc["hbm_bytes_per_launch"]
                mfma_busy = pmc.get("encoder_gemm_mfma_busy")
                break
        except (OSError, KeyError, ValueError):
            pass
    # secondary kernels, timed sparsely with HIP events on the same stream during the same timed steps
    secondary = []
    sp = prof["self_decode"]
    if sp["timed"]:
        us = 1e3 * sp["timed_ms"] / sp["timed"]
        t_mean = P + (float(n_new.mean()) - 1) / 2             # cached rows read per window and launch, averaged over the steps
        b = B * 2 * t_mean * d * es
        secondary.append({"kernel": "self_decode_kernel", "bound": "hbm", "achieved": b / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                          "unit": "GB/s", "frac": b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "avg_launch_us": us, "bytes_per_launch": b,
                          "launches_timed": sp["timed"], "launches": sp["launches"]})
    sp = prof["encoder"]
    if sp["timed"]:
        ms = sp["timed_ms"] / sp["timed"]
        fl = B * (2 * 3000 * d * 3 * dims.n_mels + 2 * 1500 * d * 3 * d + dims.enc_layers * (36000 * d * d + 9e6 * d))
        secondary.append({"kernel": "encoder pass (gemm_tn_duo_kernel + attention_kernel + layernorm)", "bound": "mfma",
                          "achieved": fl / (ms * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                          "frac": fl / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF, "ms": ms, "gemm_mfma_busy_pmc": mfma_busy})
    sp = prof["dtw"]
    if sp["timed"]:
        ms = sp["timed_ms"] / sp["timed"]
        n_rows = float(n_text.mean())
        b = B * n_rows * 1500 * 8.25
        secondary.append({"kernel": "dtw_kernel", "bound": "dependency chain", "achieved": b / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                          "unit": "GB/s", "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms": ms,
                          "us_per_antidiagonal": 1e3 * ms / (n_rows + 1499) if n_rows else None})
    # one whole decode step (all its launches), sampled one step in eight: its bytes are the decoder weights once, plus per
    # window the cross K/V of every layer and the self K/V cache read so far
    Ld, V = dims.dec_layers, dims.vocab
    w_dec = (Ld * 14 * d * d + V * d) * es
    kv_cross_step = windows_per_launch * Ld * 2 * 1500 * d * es
    sp = prof["decode_step"]
    step_ms = sp["timed_ms"] / sp["timed"] if sp["timed"] else None
    self_us = (1e3 * prof["self_decode"]["timed_ms"] / prof["self_decode"]["timed"]) if prof["self_decode"]["timed"] else 0.0
    if step_ms:
        kv_self_step = windows_per_launch * Ld * 2 * d * es * (P + (float(n_new.mean()) - 1) / 2)
        sb = w_dec + kv_cross_step + kv_self_step
        gemm_ms = max(1e-6, step_ms - Ld * (avg_ms + self_us * 1e-3))       # what is left of a step beside its attention kernels
        secondary.append({"kernel": "decode step (all launches of one greedy step)", "bound": "hbm", "achieved": sb / (step_ms * 1e-3) / 1e9,
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sb / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms": step_ms,
                          "bytes_per_step": sb, "weight_bytes_per_step": w_dec, "steps_timed": sp["timed"],
                          "weight_products_ms": gemm_ms,
                          "launches_per_step": 8 * Ld + (2 if (B <= 16 and args.dtype == "bf16") else 3),
                          "weight_products_achieved_GBs": w_dec / (gemm_ms * 1e-3) / 1e9})
    model_ms = path_roofline_ms(dims, B, P, args.gen, es)
    solo_ms = 1e3 * solo_elapsed / solo_steps
    ov_ms = prof_ov["timed_ms"] / max(1, prof_ov["timed"])
    overlapped = {"avg_launch_ms": ov_ms, "achieved": bytes_per_launch / (ov_ms * 1e-3) / 1e9 if ov_ms > 0 else 0.0,
                  "frac": bytes_per_launch / (ov_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ov_ms > 0 else 0.0,
                  "launches_timed": prof_ov["timed"],
                  "note": f"same kernel inside the headline region ({args.inflight} batches in flight share the GPU)"}
    # the dominant kernel by time per decode step: the cross-attention K/V stream when the batch is large, the decoder's
    # weight products (read once per step whatever the batch, one dependent launch each) when it is small or the model wide
    if not step_ms or Ld * avg_ms >= max(1e-6, step_ms - Ld * (avg_ms + self_us * 1e-3)):
        dominant = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic, "kernel": "cross_decode_kernel", "bytes_per_launch": bytes_per_launch,
                    "windows_per_launch": windows_per_launch, "avg_launch_ms": avg_ms, "launches_timed": prof["timed"],
                    "launches": prof["launches"]}
    else:
        gemm_ms = max(1e-6, step_ms - Ld * (avg_ms + self_us * 1e-3))
        a = w_dec / (gemm_ms * 1e-3) / 1e9
        dominant = {"bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS, "traffic": None,
                    "kernel": "decode-step weight products (gemm_step_ln_kernel + gemm_skinny_kernel + gemm_rows_hilo_kernel)",
                    "bytes_per_launch": w_dec, "launch": f"the {6 * Ld + 1} weight products of one decode step, taken together",
                    "avg_launch_ms": gemm_ms, "launches_timed": sp["timed"],
                    "note": "the weight products take more of a decode step than the cross-attention K/V stream of this batch: duration = a "
                            "whole decode step (HIP events, one step in eight) minus its attention decode launches (timed separately)",
                    "cross_decode_kernel": {"achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "avg_launch_ms": avg_ms,
                                            "bytes_per_launch": bytes_per_launch}}
    line = {
        "metric": f"audio-sec/sec (RTFx) whole node, whisper-{args.model} greedy + word timestamps",
        "value": value, "unit": "audio-sec/sec", "n_gpus": world,
        "ranks_seen": dist.get_world_size() if world > 1 else 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"whisper-{args.model} shapes (random-init weights), batches of {B} x 30 s synthetic windows per GPU, "
                               f"{args.inflight} batch(es) in flight per GPU, {P}-token diarization-style prompt, greedy decode of "
                               f"{args.gen} tokens per window with alignment-head capture, median-filter + DTW word alignment",
                   "global_batch": n_global, "windows_per_gpu": B, "batches_in_flight": args.inflight, "prompt_len": P, "generated_per_window": float(n_new.mean()),
                   "text_tokens_per_window": float(n_text.mean()), "parallelism": f"dp{world} (independent windows)"},
        "roofline": dict(dominant, **{
                     "region": f"{solo_steps} timed steps of this run with one batch in flight ({solo_ms:.1f} ms per step); with "
                               "several batches in flight a kernel's duration includes the time it shares the GPU with the others",
                     "overlapped": overlapped, "secondary": secondary,
                     "path_model_ms": model_ms, "path_frac": model_ms / ms_per_step, "path_frac_one_in_flight": model_ms / solo_ms}),
        "one_in_flight": {"value": 30.0 * n_global * solo_steps / solo_elapsed, "ms_per_step": solo_ms, "steps": solo_steps},
    }
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:   # the host baseline is reported by the single-GPU run only
            line["cpu_baseline"] = cpu_baseline(dims, sd, heads, prompt, args.gen)
        print(json.dumps(line), flush=True)
    if pool:
        pool.close()
    else:
        eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
