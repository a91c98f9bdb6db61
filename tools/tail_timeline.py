#!/usr/bin/env python3
"""Timeline of a tail lock-step from a rocprofv3 kernel trace: per kernel its mean duration and the mean idle gap in front of
it (end of the previous kernel on the device -> its start), over the steady part of tools/tail_bench.py.
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python tools/tail_bench.py 8 --steps 208
    python tools/tail_timeline.py DIR"""
import csv, glob, os, sys, json, collections

def main(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
    rows.sort()
    # the steady part: the last 40% of the launches (the timed evaluations come last)
    rows = rows[int(len(rows) * 0.6):]
    dur, gap, cnt = collections.defaultdict(float), collections.defaultdict(float), collections.Counter()
    prev_end = None
    for s, e, n in rows:
        dur[n] += e - s
        if prev_end is not None:
            gap[n] += max(0, s - prev_end)
        prev_end = max(prev_end or 0, e)
        cnt[n] += 1
    out = {n: {"calls": cnt[n], "dur_us": round(dur[n] / cnt[n] / 1e3, 2), "gap_before_us": round(gap[n] / cnt[n] / 1e3, 2)}
           for n in cnt if cnt[n] >= 20}
    span = (rows[-1][1] - rows[0][0]) / 1e3
    print(json.dumps({"span_us": round(span, 1), "kernels": out}, indent=1))

if __name__ == "__main__":
    main(sys.argv[1])
