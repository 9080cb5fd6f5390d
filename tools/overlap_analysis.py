#!/usr/bin/env python3
"""What runs beside what when several batches are in flight: from a rocprofv3 --kernel-trace CSV, the share of the busy
window in which 0, 1, 2, ... kernels are executing, and for each kernel class (HBM streams, matrix-core GEMMs /
attention, step GEMMs, rest) the share in which at least one of its kernels is executing.
Usage: overlap_analysis.py <rocprof output dir> [skip_fraction]   (the first skip_fraction of the trace is warm-up)"""
import csv
import glob
import sys

d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]


def klass(name):
    if "cross_decode" in name or "self_decode" in name or "layernorm" in name or "cross_combine" in name:
        return "hbm"
    if "gemm_skinny" in name or "gemm_step_ln" in name or "gemm_rows_hilo" in name or "greedy_next" in name:
        return "step"
    if "gemm_tn" in name or "attention_kernel" in name:
        return "mfma"
    return "rest"


rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), klass(r["Kernel_Name"])) for r in csv.DictReader(open(f))]
t_lo, t_hi = min(r[0] for r in rows), max(r[1] for r in rows)
t_from = t_lo + skip * (t_hi - t_lo)
ev = []
for s, e, k in rows:
    if e <= t_from:
        continue
    s = max(s, t_from)
    ev.append((s, 1, k))
    ev.append((e, -1, k))
ev.sort()
active = {"hbm": 0, "step": 0, "mfma": 0, "rest": 0}
by_count, by_class, combos = {}, {k: 0.0 for k in active}, {}
prev = t_from
for t, dlt, k in ev:
    dt = t - prev
    if dt > 0:
        n = sum(active.values())
        by_count[n] = by_count.get(n, 0.0) + dt
        for c, v in active.items():
            if v:
                by_class[c] += dt
        key = "+".join(c for c in ("hbm", "mfma", "step", "rest") if active[c]) or "idle"
        combos[key] = combos.get(key, 0.0) + dt
    active[k] += dlt
    prev = t
total = t_hi - t_from
print(f"window {total / 1e6:.1f} ms, {len(rows)} kernels in the trace")
print("kernels executing at once:", "  ".join(f"{n}: {100 * v / total:.1f}%" for n, v in sorted(by_count.items())))
print("some kernel of the class executing:", "  ".join(f"{c}: {100 * v / total:.1f}%" for c, v in by_class.items()))
print("class combinations:", "  ".join(f"{c}: {100 * v / total:.1f}%" for c, v in sorted(combos.items(), key=lambda kv: -kv[1])))
