#!/usr/bin/env python3
"""profiles/<prefix>_pmc.json from tools/collect_pmc_regimes.sh: per regime (window shape) of the streaming fc kernel the bytes the
memory side of L2 moved per launch -- FETCH_SIZE x 2 (16-byte-per-lane streaming loads: MI355X_MICROARCH.md's correction) +
WRITE_SIZE, each from its own --pmc pass -- per member-step, next to two floors computed from the same noise indices: every pair's
slice once (an antithetic pair shares its noise) and every *distinct* table row once (the slices of a window overlap in the table).
    python tools/summarize_pmc_regimes.py gpurun_out/r03p/pmc_regimes r03"""
import csv, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
D, PFX = sys.argv[1], sys.argv[2]
REGIMES = [("full_1window", 2500, 1), ("full_3windows", 2500, 3), ("full_4windows", 2500, 4), ("half_4windows", 1250, 4), ("third_4windows", 800, 4)]
P, FCW, FC_FLOATS, NOISE = 1009058, 12432, 3872 * 256, 250_000_000


def fc_kernel(path):
    rows = [r for r in csv.DictReader(open(path)) if 'k_fc_ring' in r['kernel'] or 'k_fc_duo' in r['kernel'] or 'k_fc2' in r['kernel'] or r['kernel'].startswith('dne::k_fc<')]
    return max(rows, key=lambda r: float(r['avg_counter_KB']) * int(r['dispatches']))      # the streaming kernel of this regime


def unique_row_bytes(idx):
    """bytes of distinct noise-table floats under the fc part of the given slices (union of intervals)"""
    s = np.sort(idx + FCW)
    e = s + FC_FLOATS
    total, cur_s, cur_e = 0, s[0], e[0]
    for a, b in zip(s[1:], e[1:]):
        if a > cur_e:
            total += cur_e - cur_s; cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    return 4 * int(total + cur_e - cur_s)


out = {"regimes": [], "fetch_correction": "FETCH_SIZE x 2: 16 B/lane streaming loads, per MI355X_MICROARCH.md (the uncorrected value would be below the "
                                          "noise bytes that must be read); FETCH_SIZE counts every L2 miss, i.e. Infinity-Cache hits too -- fabric-side traffic, an upper bound on HBM bytes",
       "command": "bash tools/collect_pmc_regimes.sh <tag>  (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace, separate passes, tools/kbench.py --pairs N --reps 1 --tslimit 6 with DNE_NSUB windows)"}
for name, pairs, nsub in REGIMES:
    f, w = os.path.join(D, name + ".FETCH_SIZE.csv"), os.path.join(D, name + ".WRITE_SIZE.csv")
    if not (os.path.exists(f) and os.path.exists(w)):
        continue
    kf, kw = fc_kernel(f), fc_kernel(w)
    units = 2.0 * pairs / nsub
    fetch, write = float(kf['avg_counter_KB']) * 1024 * 2, float(kw['avg_counter_KB']) * 1024
    from dne_hip import es
    _, idx, _ = es.generation_inputs(NOISE, P, pairs, 0, 0, 1)      # kbench's indices (rep 0), ascending: window w = the w-th contiguous part
    lo = [int(pairs * s / nsub) for s in range(nsub + 1)]
    uniq = float(np.mean([unique_row_bytes(idx[lo[s]:lo[s + 1]]) for s in range(nsub)]))
    out["regimes"].append({
        "regime": "%s: %d active pairs in %d window(s), %s" % (name, pairs, nsub, "k_fc_ring: one unit per wave, eight per workgroup" if pairs >= 1500 else "k_fc_duo: one unit per wave"),
        "kernel": kf['kernel'], "grid_size": int(kf['grid_size']), "dispatches": int(kf['dispatches']), "units_per_launch": units,
        "FETCH_SIZE_KB_avg": float(kf['avg_counter_KB']), "WRITE_SIZE_KB_avg": float(kw['avg_counter_KB']),
        "hbm_bytes_per_launch": fetch + write, "hbm_bytes_per_unit": (fetch + write) / units,
        "algorithmic_bytes_per_unit": 4064456,
        "floor_pair_sharing_bytes_per_unit": 4.0 * FC_FLOATS / 2 + 28224,          # each pair's fc noise once for its two members + the u8 stack
        "floor_unique_rows_bytes_per_unit": uniq / units,                            # every distinct table row under the window's slices once
        "moved_over_unique_rows": (fetch + write) / uniq})
if out["regimes"]:
    out["k_fc_step"] = {k: out["regimes"][0][k] for k in ("kernel", "grid_size", "units_per_launch", "FETCH_SIZE_KB_avg", "WRITE_SIZE_KB_avg",
                                                         "hbm_bytes_per_launch", "hbm_bytes_per_unit", "algorithmic_bytes_per_unit")}
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_pmc.json" % PFX), "w"), indent=1)
for r in out["regimes"]:
    print("%-70s %6.0f units  %.2f MB/unit  (pair floor %.2f, unique-row floor %.2f MB/unit; moved / unique rows %.1fx)" % (
        r["regime"][:70], r["units_per_launch"], r["hbm_bytes_per_unit"] / 1e6, r["floor_pair_sharing_bytes_per_unit"] / 1e6,
        r["floor_unique_rows_bytes_per_unit"] / 1e6, r["moved_over_unique_rows"]))
