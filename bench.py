#!/usr/bin/env python3
"""Benchmark of the transcription hot path on MI355X.

One "step" = one pass of the whole path (log-mel -> encoder -> cross K/V -> KV-cached greedy decode with
attention-row capture -> teacher-forced row fix-up -> cost matrix -> DTW) over one batch of synthetic 30 s
windows per GPU, inputs resident in HBM.  Metric: audio-seconds per second (RTFx), whole job.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Weak scaling: every rank transcribes its own `--batch` windows (independent windows shard with no data-path
collective); the only exchange is the final all_gather of token / boundary tensors (RCCL).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SOT, EOT, LANG_EN, TRANSCRIBE, TS_BEGIN, TS_END, SOT_PREV = 50258, 50257, 50259, 50359, 50364, 51864, 50361


def make_prompt(window_idx: int, n_prompt_ids: int = 160, n_prefix: int = 5):
    """[sot_prev] + fixed pseudo prompt ids + [sot, lang, transcribe, <|0.00|>] + prefix ids: stands in for the
    tokenised SPEAKERS(...) prompt and a ' [SPEAKER 1]' prefix (core.py:874-922).  170 tokens."""
    rng = np.random.default_rng(99)          # the same prompt text for every window
    body = rng.integers(1000, 40000, size=n_prompt_ids).tolist()
    prefix = rng.integers(1000, 40000, size=n_prefix).tolist()
    return [SOT_PREV] + body + [SOT, LANG_EN, TRANSCRIBE, TS_BEGIN] + prefix


def synth_audio_gpu(B: int, seed: int, device) -> torch.Tensor:
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    noise = 0.1 * torch.randn((B, 480000), generator=g, device=device, dtype=torch.float32)
    t = torch.arange(480000, device=device, dtype=torch.float32) / 16000.0
    chirp = 0.3 * torch.sin(2 * np.pi * (200 * t + (2800 / 60.0) * t * t))
    return (noise + chirp[None]).contiguous()


def cpu_baseline(dims, sd, heads, prompt, n_gen: int, n_sample_steps: int = 48):
    """The oracle (CPU restatement of the reference's schedule) on ONE window of the same workload, bounded:
    log-mel + encoder once, `n_sample_steps` of the n_gen cache-less greedy passes and as many of the n_gen
    attention-replay passes (prefix lengths spread evenly over the real range), extrapolated linearly."""
    from oracle import whisper_oracle as O
    om = O.OracleModel(sd, dims.n_heads)
    rng = np.random.default_rng(1234)
    audio = (0.1 * rng.standard_normal(480000)).astype(np.float32)
    t0 = time.perf_counter()
    mel = O.log_mel_segment(audio, dims.n_mels)
    enc = om.encode(mel)
    t_front = time.perf_counter() - t0
    P = len(prompt)
    lens = np.linspace(P, P + n_gen - 1, n_sample_steps).round().astype(int)
    filler = rng.integers(1000, 40000, size=P + n_gen).tolist()
    t_logits = t_attn = 0.0
    for T in lens:
        ids = (list(prompt) + filler)[:T]
        t0 = time.perf_counter()
        lg = om.decode_full(ids, enc)
        int(np.argmax(lg[0, -1]))
        t_logits += time.perf_counter() - t0
        t0 = time.perf_counter()
        om.decode_full(ids, enc, want_attn=True)
        t_attn += time.perf_counter() - t0
    t0 = time.perf_counter()
    att = [np.random.default_rng(5).random((1, dims.n_heads, n_gen, 1500), dtype=np.float32) for _ in range(dims.dec_layers)]
    cost = O.cost_matrix(att, heads)
    t_cost = time.perf_counter() - t0
    t_dtw = 0.0
    try:
        import ctypes
        so = os.path.join(ROOT, "oracle", "_build", "libdtw_ref.so")
        lib = ctypes.CDLL(so)
        n, m = cost.shape
        pi = np.zeros(n + m, np.int32); pj = np.zeros(n + m, np.int32); bd = np.zeros(n + 1, np.int32)
        t0 = time.perf_counter()
        lib.dtw_symmetric1_ref(cost.ctypes.data_as(ctypes.c_void_p), n, m, pi.ctypes.data_as(ctypes.c_void_p),
                               pj.ctypes.data_as(ctypes.c_void_p), bd.ctypes.data_as(ctypes.c_void_p), 1500)
        t_dtw = time.perf_counter() - t0
    except OSError:
        pass
    scale = n_gen / n_sample_steps
    total = t_front + scale * (t_logits + t_attn) + t_cost + t_dtw
    return {"value": 30.0 / total, "unit": "audio-sec/sec", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": (f"1 window of the same config on the host CPU with the reference's cache-less schedule: log-mel + encoder "
                       f"timed once ({t_front:.2f}s), {n_sample_steps} of {n_gen} greedy decoder passes ({t_logits:.2f}s) and "
                       f"{n_sample_steps} of {n_gen} attention-replay passes ({t_attn:.2f}s) at prefix lengths {lens.tolist()}, "
                       f"scaled x{scale:.0f}; cost matrix {t_cost:.3f}s, DTW {t_dtw:.3f}s; torch fp32, {torch.get_num_threads()} threads")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="small")
    ap.add_argument("--batch", type=int, default=128, help="windows per GPU")
    ap.add_argument("--gen", type=int, default=128, help="tokens generated per window")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo only to rehearse N > 1 on one GPU)")
    ap.add_argument("--device", type=int, default=None, help="GPU index for every rank (rehearsal only; default LOCAL_RANK)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    if args.device is not None:
        local = args.device
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    from whisper_amd import engine as E
    from whisper_amd.config import alignment_heads, alignment_heads_key, dims_for
    from whisper_amd.parallel import gather_results
    from whisper_amd.weights import synthetic_state_dict

    dims = dims_for(args.model)
    sd = synthetic_state_dict(dims, seed=0)
    heads = alignment_heads(alignment_heads_key(args.model, dims.d_model, "en"), dims.dec_layers, dims.n_heads)
    B = args.batch
    eng = E.Engine(dims, sd, dtype=args.dtype, max_batch=B, align_heads=heads, device=f"cuda:{local}")
    prompt = make_prompt(0)
    P = len(prompt)
    max_tokens = P + args.gen
    prompts = [prompt] * B
    audio = synth_audio_gpu(B, 1234 + rank, device)
    n_samples = torch.full((B,), 480000, dtype=torch.int32, device=device)

    def step():
        res = eng.transcribe_windows(audio, n_samples, prompts, max_tokens, EOT, SOT, TS_BEGIN, TS_END)
        if world > 1:   # the path's only exchange: gather tokens and boundary frames on every rank
            small = {k: res[k] for k in ("new_tokens", "boundaries", "n_new", "n_text")}
            gather_results(small, B * world, rank, world)
        return res

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = step()
    fence()
    eng.profile_begin(16384)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_end()
    t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    n_new = res["n_new"].cpu().numpy()
    n_text = res["n_text"].cpu().numpy()
    audio_seconds = 30.0 * B * world * args.steps
    value = audio_seconds / elapsed
    es = 2 if args.dtype == "bf16" else 4
    # algorithmic bytes of ONE cross-attention decode launch: K and V of one decoder layer for the windows it
    # covers (the decode loop runs disjoint window ranges on separate streams, so a launch covers B / n_streams)
    windows_per_launch = prof["windows"] / max(1, prof["launches"])
    bytes_per_launch = windows_per_launch * 2 * 1500 * dims.d_model * es
    avg_ms = prof["timed_ms"] / max(1, prof["timed"])
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM bytes per launch of the same kernel from the PMC counters (separate rocprofv3 --pmc passes, summary
    # committed under profiles/); only quoted when it was measured on this very configuration
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_cross_decode.json")) as fh:
            pmc = json.load(fh)
        if (pmc["model"], pmc["batch"], pmc["dtype"]) == (args.model, B, args.dtype) and abs(windows_per_launch - B) < 1e-9:
            traffic = pmc["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    line = {
        "metric": "audio-sec/sec (RTFx) whole node, whisper-small greedy + word timestamps",
        "value": value, "unit": "audio-sec/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"whisper-{args.model} shapes (random-init weights), {B} x 30 s synthetic windows per GPU, "
                               f"{P}-token diarization-style prompt, greedy decode of {args.gen} tokens per window with "
                               "alignment-head capture, median-filter + DTW word alignment",
                   "global_batch": B * world, "prompt_len": P, "generated_per_window": float(n_new.mean()),
                   "text_tokens_per_window": float(n_text.mean()), "parallelism": f"dp{world} (independent windows)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                     "traffic": traffic, "kernel": "cross_decode_kernel", "bytes_per_launch": bytes_per_launch, "windows_per_launch": windows_per_launch,
                     "avg_launch_ms": avg_ms, "launches_timed": prof["timed"], "launches": prof["launches"]},
    }
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:   # the host baseline is reported by the single-GPU run only
            line["cpu_baseline"] = cpu_baseline(dims, sd, heads, prompt, args.gen)
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
