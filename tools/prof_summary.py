#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace CSV: per kernel (and grid shape for GEMMs) calls / total / average."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
passes = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
g = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    for a, b in (("_ZN12_GLOBAL__N_1", ""),):
        short = short.replace(a, b)
    key = short[:60]
    if "gemm" in n:
        key += f" grid={int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']}"
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    g[key][0] += 1
    g[key][1] += dur
tot = sum(v[1] for v in g.values())
print(f"total kernel time {tot / 1e3 / passes:.2f} ms per pass ({passes:g} passes in trace)")
for k, v in sorted(g.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{k:80s} calls/pass {v[0] / passes:8.1f}  ms/pass {v[1] / 1e3 / passes:8.2f}  avg us {v[1] / v[0]:9.1f}  {100 * v[1] / tot:5.1f}%")
