#!/usr/bin/env python3
"""Fold tools/collect_pmc_duo_mem.sh's passes into profiles/<prefix>_pmc_fc_duo_mem.json: the vector-memory path's counters of
k_fc_duo (TA / TCP / TCC / fabric side), alone at 2500 pairs in one window and on bench.py's own launch mix, per dispatch and as the
ratios that say where the kernel waits:
  ta_busy_frac             TA_TA_BUSY summed over the 256 CUs' address units / (256 x the kernel's GRBM_GUI_ACTIVE cycles)
  tcp_*_stall_frac         the L1's stall cycles over the same denominator
  l1_hit_frac              1 - TCP_TCC_READ_REQ x 128 B / bytes the waves asked for (TA_FLAT_READ_WAVEFRONTS x 1 KiB)
  l2_hit_frac              TCC_HIT / (TCC_HIT + TCC_MISS)
  l2_read_req_latency_cyc  TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ (cycles an L2 read request of the L1 is outstanding)
  fabric_bytes             TCC_EA0_RDREQ_32B x 32 + (TCC_EA0_RDREQ - TCC_EA0_RDREQ_32B) x 128
    python tools/summarize_pmc_duo_mem.py gpurun_out/<tag>/pmc_duo_mem r05"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D, PFX = sys.argv[1], sys.argv[2]
TAG = os.environ.get("KTAG", "k_fc_ring")   # the streaming fc kernel of the run (KTAG=k_fc_duo for DNE_FC_RING=0 collections)
N_CU, N_TCC = 256, 128   # address units / L1s; L2 channels (16 per XCD)


def table(path):
    out = {}
    for r in csv.DictReader(open(path)):
        if TAG in r["kernel"]:
            k = out.setdefault(r["counter"], {"dispatches": 0, "sum": 0.0})
            k["dispatches"] += int(r["dispatches"]); k["sum"] += float(r["sum"])
            out["_kernel"] = r["kernel"]
    return out


def kernel_ns(prefix):
    p = os.path.join(D, prefix + ".kernel_ns.csv")
    if not os.path.exists(p):
        return None
    for r in csv.DictReader(open(p)):
        if TAG in r["kernel"]:
            return {"dispatches": int(r["dispatches"]), "total_ns": int(r["total_ns"])}
    return None


def bench_units(prefix):
    for p in sorted(glob.glob(os.path.join(D, prefix + ".*.json"))):
        try:
            line = [l for l in open(p) if l.startswith("{")][-1]
            ag = json.loads(line)["roofline"]["all_generations"]
            return ag["units"], ag["launches"]
        except Exception:
            continue
    return None, None


doc = {"kernel": None, "note": "one rocprofv3 --pmc pass per counter group (tools/collect_pmc_duo_mem.sh); counter collection serialises the dispatches, so "
                               "every launch is measured with the chip to itself; sums over all dispatches of the kernel in the pass"}
for prefix in ("alone", "mix"):
    c = {}
    for p in sorted(glob.glob(os.path.join(D, prefix + ".*.csv"))):
        if p.endswith("kernel_ns.csv"):
            continue
        t = table(p)
        doc["kernel"] = t.pop("_kernel", doc["kernel"])
        c.update(t)
    if not c:
        continue
    n = max(v["dispatches"] for v in c.values())
    if prefix == "alone":
        units = 5000.0 * n
    else:
        units, launches = bench_units(prefix)
    S = lambda k: c[k]["sum"] if k in c else None
    r = {"dispatches": n, "member_steps": units, "counters_sum": {k: v["sum"] for k, v in sorted(c.items())}}
    kn = kernel_ns(prefix)
    if kn:
        r["kernel_ms_per_dispatch_under_counter_collection"] = kn["total_ns"] / kn["dispatches"] * 1e-6
    d = {}
    gui = S("GRBM_GUI_ACTIVE")
    if gui:
        cyc = gui / 8.0   # GRBM_GUI_ACTIVE is reported per XCD and summed over the eight: shader cycles of the dispatches
        d["cycles_per_dispatch"] = cyc / n
        for k, name, inst in (("TA_TA_BUSY_sum", "ta_busy_frac", N_CU), ("TA_ADDR_STALLED_BY_TC_CYCLES_sum", "ta_addr_stalled_by_tc_frac", N_CU),
                              ("TA_DATA_STALLED_BY_TC_CYCLES_sum", "ta_data_stalled_by_tc_frac", N_CU),
                              ("TCP_PENDING_STALL_CYCLES_sum", "tcp_pending_stall_frac", N_CU), ("TCP_TCR_TCP_STALL_CYCLES_sum", "tcp_tcr_stall_frac", N_CU),
                              ("TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", "tcp_read_tagconflict_stall_frac", N_CU),
                              ("TCP_TCP_TA_DATA_STALL_CYCLES_sum", "tcp_ta_data_stall_frac", N_CU),
                              ("TCP_GATE_EN1_sum", "tcp_gate_en1_frac", N_CU), ("TCP_GATE_EN2_sum", "tcp_gate_en2_frac", N_CU),
                              ("TCC_TAG_STALL_sum", "tcc_tag_stall_frac", N_TCC), ("TCC_BUSY_sum", "tcc_busy_frac", N_TCC)):
            if S(k) is not None:
                d[name] = S(k) / (inst * cyc)
    if S("TA_FLAT_READ_WAVEFRONTS_sum"):
        asked = S("TA_FLAT_READ_WAVEFRONTS_sum") * 1024.0   # nearly all of the kernel's vector loads are 16 bytes per lane
        d["bytes_asked_per_unit"] = asked / units if units else None
        if S("TCP_TCC_READ_REQ_sum"):   # an L2 request of the L1 is a 128-byte line (TCC_EA0_RDREQ x 128 B reproduces FETCH_SIZE x 2)
            d["l2_read_req_bytes_per_unit"] = S("TCP_TCC_READ_REQ_sum") * 128.0 / units if units else None
            d["l1_hit_frac"] = 1.0 - S("TCP_TCC_READ_REQ_sum") * 128.0 / asked
        if S("TA_TA_BUSY_sum") and gui:
            d["ta_busy_cycles_per_wave_load"] = S("TA_TA_BUSY_sum") / S("TA_FLAT_READ_WAVEFRONTS_sum")
    if S("TCP_TCC_READ_REQ_sum") and S("TCP_TCC_READ_REQ_LATENCY_sum"):
        d["l2_read_req_latency_cyc"] = S("TCP_TCC_READ_REQ_LATENCY_sum") / S("TCP_TCC_READ_REQ_sum")
    if S("TCP_TOTAL_CACHE_ACCESSES_sum") and S("TCP_TCP_LATENCY_sum"):
        d["l1_access_latency_cyc"] = S("TCP_TCP_LATENCY_sum") / S("TCP_TOTAL_CACHE_ACCESSES_sum")
    if S("TCC_HIT_sum") is not None and S("TCC_MISS_sum") is not None and S("TCC_HIT_sum") + S("TCC_MISS_sum") > 0:
        d["l2_hit_frac"] = S("TCC_HIT_sum") / (S("TCC_HIT_sum") + S("TCC_MISS_sum"))
        d["l2_requests_per_unit"] = S("TCC_REQ_sum") / units if S("TCC_REQ_sum") and units else None
    if S("TCC_EA0_RDREQ_sum") is not None:
        b32 = S("TCC_EA0_RDREQ_32B_sum") or 0.0
        d["fabric_read_bytes_per_unit"] = (b32 * 32.0 + (S("TCC_EA0_RDREQ_sum") - b32) * 128.0) / units if units else None   # (x 128 B: the gfx950 correction of MI355X_MICROARCH.md)
        if S("TCC_EA0_RDREQ_DRAM_sum") is not None:
            d["fabric_reads_to_dram_frac"] = S("TCC_EA0_RDREQ_DRAM_sum") / max(S("TCC_EA0_RDREQ_sum"), 1.0)
    if S("SQ_WAVE_CYCLES"):
        for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
            if S(k) is not None:
                d[k + "_over_wave_cycles"] = S(k) / S("SQ_WAVE_CYCLES")
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT"):
            if S(k) is not None and units:
                d[k + "_per_unit"] = S(k) / units
        if S("SQ_ACTIVE_INST_VALU") is not None and gui:
            d["valu_busy_frac_of_simd_time"] = S("SQ_ACTIVE_INST_VALU") * 4.0 / (1024.0 * gui / 8.0)
    r["derived"] = d
    doc[prefix] = r
out = os.path.join(ROOT, "profiles", "%s_pmc_%s_mem.json" % (PFX, TAG.replace("k_", "")))
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps({k: v.get("derived") for k, v in doc.items() if isinstance(v, dict)}, indent=1))
print("wrote", out)
