"""Development probe: do two independent batches overlap on one GPU?  Starts N child processes on GPU 0, each with its own
context and batch, child r delayed by r * offset ms, no collectives; prints each child's pass times and the aggregate
rate over the window in which all children were running.  Usage: python tools/overlap_probe.py [--procs 2] [--batch 128]
[--offset-ms 160] [--steps 6]"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(args):
    sys.path.insert(0, ROOT)
    import torch
    import bench as Bn
    from whisper_amd import engine as E
    from whisper_amd.config import alignment_heads, alignment_heads_key, dims_for
    from whisper_amd.weights import synthetic_state_dict
    dims = dims_for(args.model)
    sd = synthetic_state_dict(dims, seed=0)
    heads = alignment_heads(alignment_heads_key(args.model, dims.d_model, "en"), dims.dec_layers, dims.n_heads)
    B = args.batch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    eng = E.Engine(dims, sd, dtype="bf16", max_batch=B, align_heads=heads, device="cuda:0")
    prompt = Bn.make_prompt(0)
    audio = Bn.synth_audio_gpu([1234 + w for w in range(B)], dev)
    n_samples = torch.full((B,), 480000, dtype=torch.int32, device=dev)
    run = lambda: eng.transcribe_windows(audio, n_samples, [prompt] * B, len(prompt) + 128, Bn.EOT, Bn.SOT, Bn.TS_BEGIN, Bn.TS_END)
    run()
    torch.cuda.synchronize()
    # all children start from a common wall-clock instant handed down by the parent
    while time.time() < args.start_at + args.rank * args.offset_ms * 1e-3:
        time.sleep(0.0005)
    marks = [time.time()]
    for _ in range(args.steps):
        run()
        torch.cuda.synchronize()
        marks.append(time.time())
    print(json.dumps({"rank": args.rank, "marks": marks}), flush=True)
    eng.close()


def threads_mode(args):
    """Same question inside ONE process: a context, a non-blocking stream and a host thread per batch."""
    sys.path.insert(0, ROOT)
    import threading
    import torch
    import bench as Bn
    from whisper_amd import engine as E
    from whisper_amd.config import alignment_heads, alignment_heads_key, dims_for
    from whisper_amd.weights import synthetic_state_dict
    dims = dims_for(args.model)
    sd = synthetic_state_dict(dims, seed=0)
    heads = alignment_heads(alignment_heads_key(args.model, dims.d_model, "en"), dims.dec_layers, dims.n_heads)
    B = args.batch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    prompt = Bn.make_prompt(0)
    audio = Bn.synth_audio_gpu([1234 + w for w in range(B)], dev)
    n_samples = torch.full((B,), 480000, dtype=torch.int32, device=dev)
    engs = [E.Engine(dims, sd, dtype="bf16", max_batch=B, align_heads=heads, device="cuda:0") for _ in range(args.procs)]
    streams = [torch.cuda.Stream() for _ in range(args.procs)]
    marks = [[] for _ in range(args.procs)]

    def work(r):
        torch.cuda.set_device(0)
        with torch.cuda.stream(streams[r]):
            run = lambda: engs[r].transcribe_windows(audio, n_samples, [prompt] * B, len(prompt) + 128, Bn.EOT, Bn.SOT, Bn.TS_BEGIN, Bn.TS_END)
            run()
            streams[r].synchronize()
            bar.wait()
            time.sleep(r * args.offset_ms * 1e-3)
            marks[r].append(time.time())
            for _ in range(args.steps):
                run()
                streams[r].synchronize()
                marks[r].append(time.time())

    bar = threading.Barrier(args.procs)
    torch.cuda.synchronize()
    th = [threading.Thread(target=work, args=(r,)) for r in range(args.procs)]
    for t in th: t.start()
    for t in th: t.join()
    report([{"rank": r, "marks": marks[r]} for r in range(args.procs)], args)
    for e in engs: e.close()


def report(outs, args):
    for o in outs:
        m = o["marks"]
        print("rank", o["rank"], "pass ms:", " ".join(f"{1e3 * (b - a):.1f}" for a, b in zip(m, m[1:])))
    lo = max(o["marks"][0] for o in outs)
    hi = min(o["marks"][-1] for o in outs)
    passes = 0.0
    for o in outs:   # passes (fractional at the edges) each child completed inside the common interval
        m = o["marks"]
        for a, b in zip(m, m[1:]):
            ov = max(0.0, min(b, hi) - max(a, lo))
            passes += ov / (b - a)
    print(f"common interval {1e3 * (hi - lo):.1f} ms, {passes:.2f} passes -> {1e3 * (hi - lo) / passes:.1f} ms per pass of {args.batch} windows, "
          f"RTFx {30.0 * args.batch * passes / (hi - lo):.0f}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", action="store_true")
    ap.add_argument("--procs", type=int, default=2)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--model", default="small")
    ap.add_argument("--offset-ms", type=float, default=160.0)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--rank", type=int, default=-1)
    ap.add_argument("--start-at", type=float, default=0.0)
    args = ap.parse_args()
    if args.rank >= 0:
        return child(args)
    if args.threads:
        return threads_mode(args)
    start_at = time.time() + 75.0          # children need about a minute to import torch and build their contexts
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(r), "--start-at", repr(start_at)] + sys.argv[1:],
                              stdout=subprocess.PIPE, text=True) for r in range(args.procs)]
    outs = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in procs]
    report(outs, args)


if __name__ == "__main__":
    main()
