#!/usr/bin/env python3
"""SQ / SPI / traffic counters of the full-width lock-step's kernels (tools/calls/r06_mid.sh, passes pmc_A .. pmc_G over
`tools/kbench.py --pairs 2500 --reps 1 --tslimit 6`, default four windows) -> profiles/r06_pmc_lockstep_kernels.json.
rocprofv3 --pmc SERIALISES dispatches: every figure here is the kernel with the chip to itself on a quarter of the members (1250 per
dispatch); what changes beside the streaming fc is tools/wg_clock.py's business (profiles/r06_wg_clock_*.jsonl).
Derived per kernel: waves per dispatch, VALU instructions per wave, VALU / MFMA pipe utilisation = busy quad-cycles x 4 / (CU-busy cycles x 4 SIMDs),
share of wave time spent waiting on an instruction / on LDS, LDS bank-conflict cycles per CU-busy cycle, dispatcher stalls (SPI_RA_*), HBM bytes.
    python tools/summarize_pmc_lockstep.py gpurun_out/<tag> > profiles/r06_pmc_lockstep_kernels.json"""
import csv, glob, json, os, sys
D = sys.argv[1]
KERNELS = ("k_conv12<true>", "k_fc_ring<true, 8>", "k_out<2, true>", "k_env_render")
c = {}
for p in sorted(glob.glob(os.path.join(D, "pmc_*.csv"))):
    for r in csv.DictReader(open(p)):
        k = r["kernel"].replace("dne::", "")
        if k in KERNELS:
            c.setdefault(k, {})[r["counter"]] = (float(r["sum"]), int(r["dispatches"]))
out = {"source": "rocprofv3 --pmc (7 separate passes, --kernel-trace only) over tools/kbench.py --pairs 2500 --reps 1 --tslimit 6; dispatches are serialised by the profiler: "
                 "each kernel ALONE on 1250 members per dispatch (k_env_render: all its dispatches of the run, incl. the reset's)",
       "units": "SQ_WAVE_CYCLES, SQ_ACTIVE_INST_*, SQ_WAIT_* are quad-cycles (x4 = cycles); SQ_BUSY_CU_CYCLES, SQ_VALU_MFMA_BUSY_CYCLES, SPI_RA_* are cycles; FETCH_SIZE / WRITE_SIZE KB (FETCH x2: gfx950 correction for 16 B/lane loads applies to the streaming kernel only)",
       "kernels": {}}
for k, v in c.items():
    g = lambda n: v.get(n, (0.0, 1))[0]
    nd = v["SQ_WAVES"][1]
    cu = max(g("SQ_BUSY_CU_CYCLES"), 1.0)
    waves = max(g("SQ_WAVES"), 1.0)
    wc = max(g("SQ_WAVE_CYCLES"), 1.0)
    out["kernels"][k] = {
        "dispatches": nd, "waves_per_dispatch": waves / nd, "cu_busy_cycles_per_dispatch_per_cu": cu / nd / 256,
        "waves_resident_per_cu": 4 * wc / cu, "valu_insts_per_wave": g("SQ_INSTS_VALU") / waves, "mfma_insts_per_wave": g("SQ_INSTS_MFMA") / waves,
        "valu_pipe_util": 4 * g("SQ_ACTIVE_INST_VALU") / (4 * cu), "mfma_pipe_util": g("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * cu),
        "wave_time_waiting_on_any_inst": g("SQ_WAIT_INST_ANY") / wc, "wave_time_waiting_on_lds_inst": g("SQ_WAIT_INST_LDS") / wc,
        "wave_time_issuing_any_inst": g("SQ_ACTIVE_INST_ANY") / wc, "lds_inst_quad_cycles_per_cu_cycle": 4 * g("SQ_ACTIVE_INST_LDS") / cu,
        "lds_bank_conflict_cycles_per_cu_busy_cycle": g("SQ_LDS_BANK_CONFLICT") / cu,
        "spi_stall_cycles_per_dispatch": {"lds_cu_full": g("SPI_RA_LDS_CU_FULL_CSN") / nd, "vgpr_simd_full": g("SPI_RA_VGPR_SIMD_FULL_CSN") / nd,
                                          "wave_simd_full": g("SPI_RA_WAVE_SIMD_FULL_CSN") / nd, "req_no_alloc": g("SPI_RA_REQ_NO_ALLOC_CSN") / nd},
        "hbm_bytes_per_dispatch": {"fetch_raw": g("FETCH_SIZE") * 1024 / nd, "write": g("WRITE_SIZE") * 1024 / nd},
        "raw": {n: x[0] for n, x in sorted(v.items())}}
print(json.dumps(out, indent=1))
