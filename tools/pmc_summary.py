#!/usr/bin/env python3
"""Per-kernel mean of rocprofv3 --pmc counters.  Usage: pmc_summary.py <dir> [substring filter]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    if flt and flt not in name:
        continue
    key = f"{name} grid={r['Grid_Size']}"
    a = acc[key][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
for k, cs in sorted(acc.items(), key=lambda kv: -sum(v[1] for v in kv[1].values()))[:40]:
    print(k)
    for c, (n, tot) in cs.items():
        print(f"    {c:32s} dispatches {n:6d}  mean {tot / n:16.1f}")
