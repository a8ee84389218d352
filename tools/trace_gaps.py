#!/usr/bin/env python3
"""Per-kernel duration and the idle gap in front of each kernel, from a rocprofv3 --kernel-trace CSV.
usage: tools/trace_gaps.py <dir given to rocprofv3 -d> [last N kernels]"""
import collections
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void nttx::", "")))
rows.sort()
last = int(sys.argv[2]) if len(sys.argv) > 2 else 600
rows = rows[-last:]
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
for i, (s, e, name) in enumerate(rows):
    dur[name].append(e - s)
    if i:
        gap[name].append(s - rows[i - 1][1])
for name in dur:
    d, g = dur[name], gap[name] or [0]
    print(f"{name[:90]:90s} n={len(d):4d}  dur avg {sum(d)/len(d)/1e3:7.2f} us  min {min(d)/1e3:7.2f}   gap before: avg {sum(g)/len(g)/1e3:6.2f} us  min {min(g)/1e3:6.2f}")
span = rows[-1][1] - rows[0][0]
print(f"span {span/1e3:.1f} us over {len(rows)} kernels; busy {sum(e - s for s, e, _ in rows)/1e3:.1f} us")
