#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV of a fixed-width run (tools/kbench.py / tools/ga_kbench.py: nobody dies, every lock-step has the same
width) -> what a window's chain of launches looks like: per kernel its duration AND the idle gap on its stream in front of it,
the period of a window's lock-step, how many chains are in flight (sum of durations / span), which stream sits on which hardware queue.
Only the lock-steps are summarised: everything up to the last reference-pass / set-up kernel is dropped.
    python tools/trace_summary.py <kernel_trace.csv> <label> [--csv out.csv] > summary.json"""
import csv, json, statistics, sys

path, label = sys.argv[1], sys.argv[2]
out_csv = sys.argv[sys.argv.index("--csv") + 1] if "--csv" in sys.argv else None
LOCK = ("k_conv12", "k_conv1<", "k_conv1 ", "k_conv2<", "k_fc<", "k_fc2", "k_fc_ring", "k_fc_duo", "k_fc_sub", "k_fc_tail", "k_fc_quad", "k_fc_cols", "k_unit_order", "k_out<",
        "k_env_logic", "k_env_render", "k_tail_step", "k_tail_select", "k_y2_activate", "k_compact", "k_lconv", "k_lfc", "k_lout", "k_conv1_spec", "k_conv2_spec")
rows = []
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dne::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Stream_Id", ""), r.get("Queue_Id", "")))
rows.sort()
is_lock = [(r[2] + " ").startswith(LOCK) or r[2] == "k_conv1" for r in rows]
# the LAST evaluation's lock-steps: from the first lock-step kernel behind the last k_iota (eval_core builds the active list with it) to the last one
iota = max([i for i, r in enumerate(rows) if r[2].startswith("k_iota")], default=-1)
idx = [i for i in range(iota + 1, len(rows)) if is_lock[i]]
if not idx:
    raise SystemExit("no lock-step kernels behind the last k_iota")
ls = [rows[i] for i in range(idx[0], idx[-1] + 1) if is_lock[i]]
t0 = ls[0][0]
span_us = (max(r[1] for r in ls) - t0) / 1e3
streams = {}
for a, b, k, st, q in ls:
    streams.setdefault(st or q, []).append((a, b, k, q))
per_kernel = {}
chains = {}
for st, seq in streams.items():
    prev_end = None
    starts = []
    for a, b, k, q in seq:
        e = per_kernel.setdefault(k, {"dur": [], "gap": []})
        e["dur"].append((b - a) / 1e3)
        if prev_end is not None:
            e["gap"].append(max(a - prev_end, 0) / 1e3)
        prev_end = b
    # a window's period: distance between successive launches of the first kernel of its chain (the most frequent first name)
    first = seq[0][2]
    starts = [a for a, b, k, q in seq if k == first]
    per = [(y - x) / 1e3 for x, y in zip(starts, starts[1:])]
    chains[st] = {"queue": seq[0][3], "launches": len(seq), "first_kernel": first, "period_us_median": round(statistics.median(per), 1) if per else None,
                  "busy_frac": round(sum(b - a for a, b, k, q in seq) / 1e3 / max((seq[-1][1] - seq[0][0]) / 1e3, 1e-9), 3)}
summary = {"label": label, "source": "rocprofv3 --kernel-trace", "lock_step_span_us": round(span_us, 1), "streams": chains,
           "sum_of_durations_over_span": round(sum(b - a for a, b, k, st, q in ls) / 1e3 / span_us, 2), "kernels": {}}
for k, e in sorted(per_kernel.items(), key=lambda kv: -sum(kv[1]["dur"])):
    summary["kernels"][k[:48]] = {"launches": len(e["dur"]), "dur_us_mean": round(statistics.mean(e["dur"]), 1), "dur_us_median": round(statistics.median(e["dur"]), 1),
                                  "gap_before_us_mean": round(statistics.mean(e["gap"]), 1) if e["gap"] else None,
                                  "gap_before_us_median": round(statistics.median(e["gap"]), 1) if e["gap"] else None}
print(json.dumps(summary))
if out_csv:
    with open(out_csv, "w") as f:
        f.write("# %s\nstart_us,end_us,kernel,stream,queue\n" % label)
        for a, b, k, st, q in ls:
            f.write("%.1f,%.1f,%s,%s,%s\n" % ((a - t0) / 1e3, (b - t0) / 1e3, k[:40], st, q))
