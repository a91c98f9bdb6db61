#!/bin/bash
# What the streaming fc (k_fc_ring; DNE_FC_RING=0: k_fc_duo) waits for: the vector-memory path's own counters -- TA (address unit), TCP (per-CU L1),
# TCC (per-XCD L2) and the L2's fabric side -- one rocprofv3 --pmc pass per group (kernel trace only), over
#   alone:  the kernel with the chip to itself, 2500 pairs in one window (tools/kbench.py, DNE_NSUB=1: six launches of 5000 member-steps)
#   mix:    every k_fc_duo dispatch of bench.py's own launch mix (--steps 3 --warmup 1; counter collection serialises the dispatches)
# Groups are small on purpose (a block has few counter slots; a group that does not fit fails alone and is listed in errors.log).
#   bash tools/collect_pmc_duo_mem.sh <tag>  ->  gpurun_out/<tag>/pmc_duo_mem/ ;  python tools/summarize_pmc_duo_mem.py gpurun_out/<tag>/pmc_duo_mem r05
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG/pmc_duo_mem
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps ${MIX_STEPS:-3} --warmup 1 --no-supervisor --extra none --no-cpu-baseline"
ALONE="python $R/tools/kbench.py --pairs 2500 --reps 1 --tslimit 6"
reduce() {  # counter_collection.csv -> per (kernel, counter): dispatches, sum
  python - "$1" "$2" <<'PY'
import csv, collections, sys
tot = collections.defaultdict(float); disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = (r['Kernel_Name'].split('(')[0].replace('void ', ''), r['Counter_Name'])
    tot[k] += float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
with open(sys.argv[2], 'w') as f:
    f.write("kernel,counter,dispatches,sum\n")
    for k in sorted(tot):
        f.write('"%s",%s,%d,%.1f\n' % (k[0], k[1], len(disp[k]), tot[k]))
PY
}
pass() {  # name "counters" command...   (SKIP="TAb SQa ..." leaves groups out)
  local name=$1 ctr=$2; shift 2
  if echo " ${SKIP:-} " | grep -q " ${name#*.} "; then return; fi
  timeout 240 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$O/$name.d" -o p -- "$@" > "$O/$name.json" 2> "$O/$name.err"
  f=$(find "$O/$name.d" -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then reduce "$f" "$O/$name.csv"; else echo "no counter file for $name" >> "$O/errors.log"; tail -5 "$O/$name.err" >> "$O/errors.log"; fi
  k=$(find "$O/$name.d" -name '*kernel_trace.csv' | head -1)
  if [ -n "$k" ] && [ ! -f "$O/${name%%.*}.kernel_ns.csv" ]; then
    python - "$k" "$O/${name%%.*}.kernel_ns.csv" <<'PY'
import csv, collections, sys
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r['Kernel_Name'].split('(')[0].replace('void ', '')].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
with open(sys.argv[2], 'w') as f:
    f.write("kernel,dispatches,total_ns\n")
    for k, v in sorted(d.items()):
        f.write('"%s",%d,%d\n' % (k, len(v), sum(v)))
PY
  fi
  rm -rf "$O/$name.d"
}
groups() {  # prefix command...
  local pre=$1; shift
  pass $pre.TAa  "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "$@"
  pass $pre.TAb  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "$@"
  pass $pre.GUI  "GRBM_GUI_ACTIVE" "$@"
  pass $pre.SQa  "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES" "$@"
  pass $pre.SQb  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "$@"
  pass $pre.SQl  "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "$@"
  pass $pre.TCPa "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "$@"
  pass $pre.TCPb "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" "$@"
  if [ "${FULL:-1}" = "1" ]; then pass $pre.TCPc "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_READ_sum" "$@"; fi
  pass $pre.TCCa "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "$@"
  pass $pre.TCCb "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum" "$@"
  pass $pre.TCCc "TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" "$@"
}
DNE_NSUB=1 groups alone $ALONE
if [ "${MIX:-1}" = "1" ]; then
  pass mix.TAa  "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" $BENCH
  pass mix.GUI  "GRBM_GUI_ACTIVE" $BENCH
  pass mix.TCPa "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" $BENCH
  pass mix.TCPb "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" $BENCH
  pass mix.TCCa "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" $BENCH
  pass mix.TCCb "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum" $BENCH
fi
ls -la "$O"; cat "$O/errors.log" 2>/dev/null
