#!/usr/bin/env python3
"""kernel-trace CSV of a tools/tail_bench.py run -> for the last 200 lock-steps: mean duration of every kernel of the step and
mean gap between consecutive kernels (end -> next start).  usage: python tools/tail_trace.py <dir with *_kernel_trace.csv>"""
import csv, glob, json, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in csv.DictReader(open(f))))
rows = rows[-1400:]
dur, gap, prev = {}, {}, None
for s, e, k in rows:
    dur.setdefault(k, []).append(e - s)
    if prev is not None:
        gap.setdefault(prev[2] + " -> " + k, []).append(s - prev[1])
    prev = (s, e, k)
out = {"kernel_us": {k: [len(v), round(sum(v) / len(v) / 1e3, 2)] for k, v in dur.items()},
       "gap_us": {k: [len(v), round(sum(v) / len(v) / 1e3, 2)] for k, v in gap.items() if len(v) > 20}}
print(json.dumps(out, indent=1))
