#!/usr/bin/env python3
"""Counted traffic of the GA legs: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel trace only)
over tools/ga_bench.py and tools/ga_bench.py --large, every dispatch of the run, divided by the env-steps the run made --
tools/collect_profiles_r05.sh leaves <dir>/ga.FETCH_SIZE.csv etc. (kernel, counter, dispatches, sum) and the runs' own JSON lines.
    python tools/summarize_pmc_ga.py gpurun_out/<tag>/pmc_ga r05   ->  profiles/r05_pmc_ga.json"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D, PFX = sys.argv[1], sys.argv[2]
FC = ("k_fc<", "k_fc_sub", "k_fc_tail", "k_fc_quad", "k_fc_cols", "k_lfc")


def steps_of(path):
    n = 0
    for l in open(path):
        if not l.startswith("{"):
            continue
        d = json.loads(l)
        if "gen" in d and "env_steps" in d:
            n += d["env_steps"]
        for row in d.get("deep_chains", []):
            n += row["cold"]["env_steps"] + row["warm"]["env_steps"]
    return n


out = {"note": "FETCH_SIZE x 2 (MI355X_MICROARCH.md's gfx950 correction for 16-byte streaming loads; small kernels' loads are not all of that "
               "kind, so the total is an upper bound) + WRITE_SIZE, summed over every dispatch of the run, per env-step of the run "
               "(generation 0, the steady generations and the deep-chain generations of tools/workloads.py:ga_small / ga_large)"}
for leg in ("ga", "ga_large"):
    f, w = os.path.join(D, leg + ".FETCH_SIZE.csv"), os.path.join(D, leg + ".WRITE_SIZE.csv")
    if not (os.path.exists(f) and os.path.exists(w)):
        continue
    by = {}
    for path, scale in ((f, 2048.0), (w, 1024.0)):
        for r in csv.DictReader(open(path)):
            k = by.setdefault(r["kernel"], {"bytes": 0.0, "dispatches": 0})
            k["bytes"] += float(r["sum"]) * scale
            k["dispatches"] = max(k["dispatches"], int(r["dispatches"]))
    sf, sw = steps_of(os.path.join(D, leg + ".FETCH_SIZE.json")), steps_of(os.path.join(D, leg + ".WRITE_SIZE.json"))
    steps = max(sf, sw, 1)
    total = sum(k["bytes"] for k in by.values())
    fc = sum(k["bytes"] for n, k in by.items() if any(t in n for t in FC))
    top = sorted(by.items(), key=lambda kv: -kv[1]["bytes"])[:8]
    out[leg] = {"env_steps_fetch_pass": sf, "env_steps_write_pass": sw, "bytes_per_unit": total / steps, "fc_kernels_bytes_per_unit": fc / steps,
                "algorithmic_bytes_per_unit": 4 * (4052658 if leg == "ga_large" else 1008450) + 28224,
                "top_kernels": [{"kernel": n, "dispatches": k["dispatches"], "bytes_per_unit": k["bytes"] / steps} for n, k in top]}
    print("%s: %.2f MB counted per env-step (fc kernels %.2f MB; algorithmic %.2f MB), %d env-steps" % (
        leg, total / steps / 1e6, fc / steps / 1e6, out[leg]["algorithmic_bytes_per_unit"] / 1e6, steps))
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_pmc_ga.json" % PFX), "w"), indent=1)
