"""Development probe: how well does the encoder stage of one batch overlap with the decode stage of another?  Two contexts
on one GPU, two host threads, two streams: A repeats mel -> encoder -> cross K/V, B repeats the greedy decode loop.  Prints
the time of n repetitions of each alone and of both side by side (sum = no overlap, max = full overlap)."""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--model", default="small")
    ap.add_argument("--enc-reps", type=int, default=6)
    ap.add_argument("--dec-reps", type=int, default=2)
    ap.add_argument("--dec-priority", type=int, default=0, help="stream priority of the decode stage (-1 = high)")
    ap.add_argument("--enc-priority", type=int, default=0)
    ap.add_argument("--enc-cus", type=int, default=0, help="> 0: the encoder stage's stream may use this many CUs, the decode stage's the rest")
    ap.add_argument("--mask", choices=("low", "spread"), default="spread",
                    help="low: the encoder gets mask bits 0..n-1; spread: n/8 bits out of every 32 (whatever the bit -> XCD map is, both see all of it)")
    args = ap.parse_args()
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
    ea, eb = (E.Engine(dims, sd, dtype="bf16", max_batch=B, align_heads=heads, device="cuda:0") for _ in range(2))
    sa, sb = torch.cuda.Stream(priority=args.enc_priority), torch.cuda.Stream(priority=args.dec_priority)
    if args.enc_cus > 0:
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        n = args.enc_cus
        if args.mask == "low":
            bits = [1 if i < n else 0 for i in range(256)]
        else:
            per = n // 8
            bits = [1 if (i % 32) < per else 0 for i in range(256)]
        def stream_for(bitlist):
            words = (C.c_uint32 * 8)(*[sum(b << j for j, b in enumerate(bitlist[w * 32:(w + 1) * 32])) for w in range(8)])
            h = C.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), 8, words)
            assert rc == 0, rc
            return torch.cuda.ExternalStream(h.value)
        sa, sb = stream_for(bits), stream_for([1 - b for b in bits])
        print(f"encoder stage on {sum(bits)} CUs ({args.mask}), decode stage on {256 - sum(bits)}")
    print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")

    def enc_stage():
        mel = ea.mel(audio, n_samples)
        ea.encode(mel)
        ea.cross_kv(B)

    def dec_stage():
        eb.decode_greedy([prompt] * B, len(prompt) + 128, Bn.EOT, Bn.SOT, Bn.TS_BEGIN, Bn.TS_END)

    with torch.cuda.stream(sb):      # B needs cross K/V once
        mel = eb.mel(audio, n_samples); eb.encode(mel); eb.cross_kv(B); dec_stage()
    with torch.cuda.stream(sa):
        enc_stage()
    torch.cuda.synchronize()

    def run(fn, stream, reps, out, key):
        torch.cuda.set_device(0)
        with torch.cuda.stream(stream):
            t0 = time.time()
            for _ in range(reps):
                fn()
            stream.synchronize()
            out[key] = time.time() - t0

    out = {}
    run(enc_stage, sa, args.enc_reps, out, "enc_alone")
    run(dec_stage, sb, args.dec_reps, out, "dec_alone")
    ta = threading.Thread(target=run, args=(enc_stage, sa, args.enc_reps, out, "enc_beside"))
    tb = threading.Thread(target=run, args=(dec_stage, sb, args.dec_reps, out, "dec_beside"))
    t0 = time.time()
    ta.start(); tb.start(); ta.join(); tb.join()
    both = time.time() - t0
    print(f"encoder stage x{args.enc_reps}: alone {1e3 * out['enc_alone']:.0f} ms, beside {1e3 * out['enc_beside']:.0f} ms | "
          f"decode stage x{args.dec_reps}: alone {1e3 * out['dec_alone']:.0f} ms, beside {1e3 * out['dec_beside']:.0f} ms | "
          f"both {1e3 * both:.0f} ms (sum {1e3 * (out['enc_alone'] + out['dec_alone']):.0f}, max {1e3 * max(out['enc_alone'], out['dec_alone']):.0f})",
          flush=True)
    ea.close(); eb.close()


if __name__ == "__main__":
    main()
