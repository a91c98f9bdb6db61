#!/usr/bin/env python3
"""Fold tools/collect_pmc_bench_mix.sh's passes into profiles/<prefix>_pmc.json (created by tools/summarize_pmc_regimes.py, or here):
  regimes[0] = "bench_mix": FETCH_SIZE x 2 + WRITE_SIZE summed over every k_fc_duo dispatch of bench.py's own run, per member-step
               those launches processed (roofline.all_generations of the very run that was profiled);
  sq["k_fc_duo"]: the kernel's SQ counters per member-step, alone at full width (one window) and on the bench mix.
    python tools/summarize_pmc_bench_mix.py gpurun_out/<tag>/pmc_mix r04"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D, PFX = sys.argv[1], sys.argv[2]
TAG = os.environ.get("KTAG", "k_fc_ring")   # the roofline kernel of the run (KTAG=k_fc_duo for a DNE_FC_RING=0 collection)


def table(name):
    p = os.path.join(D, name + ".csv")
    if not os.path.exists(p):
        return {}
    out = {}
    for r in csv.DictReader(open(p)):
        if TAG in r["kernel"]:
            k = out.setdefault(r["counter"], {"kernel": r["kernel"], "dispatches": 0, "sum": 0.0})
            k["dispatches"] += int(r["dispatches"]); k["sum"] += float(r["sum"])
    return out


def bench_line(name):
    p = os.path.join(D, name + ".json")
    try:
        lines = [l for l in open(p) if l.startswith("{")]
        return json.loads(lines[-1])
    except Exception:
        return None


path = os.path.join(ROOT, "profiles", "%s_pmc.json" % PFX)
doc = json.load(open(path)) if os.path.exists(path) else {"regimes": []}
doc["regimes"] = [r for r in doc.get("regimes", []) if not r["regime"].startswith("bench_mix")]
f, w = table("mix.FETCH_SIZE"), table("mix.WRITE_SIZE")
bf, bw = bench_line("mix.FETCH_SIZE"), bench_line("mix.WRITE_SIZE")
if f and w and bf and bw:
    uf, lf = bf["roofline"]["all_generations"]["units"], bf["roofline"]["all_generations"]["launches"]
    uw, lw = bw["roofline"]["all_generations"]["units"], bw["roofline"]["all_generations"]["launches"]
    fetch = f["FETCH_SIZE"]["sum"] * 1024 * 2 / uf           # KB -> B, x2: MI355X_MICROARCH.md's gfx950 correction for 16 B/lane streaming loads
    write = w["WRITE_SIZE"]["sum"] * 1024 / uw
    fixed = [r for r in doc["regimes"] if "floor_unique_rows_bytes_per_unit" in r]
    upl = uf / max(lf, 1)
    near = min(fixed, key=lambda r: abs(r["units_per_launch"] - upl)) if fixed else None
    reg = {"regime": "bench_mix: every %s dispatch of `bench.py --steps %d --warmup %d --extra none` (%d launches, %.0f member-steps per launch on average)"
                     % (TAG, bf["steps"], bf["warmup"], lf, upl),
           "kernel": f["FETCH_SIZE"]["kernel"], "dispatches": f["FETCH_SIZE"]["dispatches"], "units_per_launch": upl,
           "dispatches_match_bench": f["FETCH_SIZE"]["dispatches"] == lf and w["WRITE_SIZE"]["dispatches"] == lw,
           "FETCH_SIZE_KB_sum": f["FETCH_SIZE"]["sum"], "WRITE_SIZE_KB_sum": w["WRITE_SIZE"]["sum"], "units_fetch_pass": uf, "units_write_pass": uw,
           "hbm_bytes_per_unit": fetch + write, "algorithmic_bytes_per_unit": 4064456,
           "floor_pair_sharing_bytes_per_unit": 4.0 * 3872 * 256 / 2 + 28224}
    if near:
        reg["floor_unique_rows_bytes_per_unit"] = near["floor_unique_rows_bytes_per_unit"]
        reg["floor_source"] = "the fixed-width regime nearest in units per launch (%s): which pairs are alive in a bench launch is not recorded" % near["regime"].split(":")[0]
        reg["moved_over_unique_rows"] = (fetch + write) / near["floor_unique_rows_bytes_per_unit"]
    doc["regimes"].insert(0, reg)
    print("bench_mix: %.3f MB per member-step (FETCH x2 %.3f + WRITE %.3f), %d launches, dispatch counts match: %s"
          % ((fetch + write) / 1e6, fetch / 1e6, write / 1e6, lf, reg["dispatches_match_bench"]))


def sq(prefix, units):
    a, b = table(prefix + ".SQ_A"), table(prefix + ".SQ_B")
    c = dict(a); c.update(b)
    if not c or not units:
        return None
    out = {k + "_per_unit": v["sum"] / units for k, v in c.items()}
    out["dispatches"] = max(v["dispatches"] for v in c.values())
    wc = c.get("SQ_WAVE_CYCLES", {}).get("sum")
    if wc:
        out["wave_cycles_split"] = {k: round(c[k]["sum"] / wc, 4) for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") if k in c}
    return out


sqd = {}
alone = sq("alone", 6 * 5000.0)      # tools/kbench.py --pairs 2500 --tslimit 6: six launches of 5000 member-steps
if alone:
    alone["regime"] = "alone, 2500 pairs in one window (tools/kbench.py, DNE_NSUB=1)"
    sqd.update(alone)
ba = bench_line("mix.SQ_A")
if ba:
    mix = sq("mix", ba["roofline"]["all_generations"]["units"])
    if mix:
        sqd["bench_mix"] = mix
if sqd:
    doc.setdefault("sq", {})[TAG] = sqd
    print("sq:", json.dumps({k: v for k, v in sqd.items() if k != "bench_mix"})[:600])
doc["bench_mix_command"] = "bash tools/collect_pmc_bench_mix.sh <tag>  (rocprofv3 --pmc <one group> --kernel-trace over bench.py --steps 3 --warmup 1 --no-supervisor --extra none --no-cpu-baseline)"
json.dump(doc, open(path, "w"), indent=1)
