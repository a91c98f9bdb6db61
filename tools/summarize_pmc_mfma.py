#!/usr/bin/env python3
"""rocprofv3 counter CSV of tools/collect_pmc_mfma.sh -> per-kernel means and the matrix-core busy fraction."""
import collections, csv, glob, json, os, sys
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[k][r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
out = {}
for k, cs in acc.items():
    if not any(t in k for t in ("conv", "fc", "bn_", "env_render")):
        continue
    per = {}
    for c, vals in cs.items():
        by = collections.defaultdict(float)
        for disp, v in vals:
            by[disp] += v          # one row per XCD / SE instance: sum them per dispatch
        per[c] = sum(by.values()) / len(by)
        n = len(by)
    e = {"dispatches": n, **{c: per[c] for c in sorted(per)}}
    if per.get("GRBM_GUI_ACTIVE"):
        e["mfma_busy_frac"] = per.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * per["GRBM_GUI_ACTIVE"] / 8)
        if per.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac"] = per.get("SQ_LDS_BANK_CONFLICT", 0.0) / per["SQ_LDS_IDX_ACTIVE"]
        if per.get("SQ_WAVE_CYCLES"):
            for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
                e[c.lower() + "_frac"] = per.get(c, 0.0) / per["SQ_WAVE_CYCLES"]
    out[k] = e
print(json.dumps({"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY "
                             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -- python tools/kbench.py --reps 2 --tslimit 2",
                  "mfma_busy_frac": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs), mean over dispatches", "kernels": out}, indent=1))
