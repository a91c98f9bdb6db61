#!/bin/bash
# HBM-side traffic and issue counters of the streaming fc kernel (k_fc_duo) on bench.py's OWN launch mix:
# rocprofv3 --pmc over the bench command itself (kernel trace only, one counter group per pass -- FETCH_SIZE and WRITE_SIZE do not fit
# one pass), summed over EVERY k_fc_duo dispatch and divided by the member-steps those launches processed (bench.py's
# roofline.all_generations, warm-up included).  Plus the kernel's SQ counters alone at full width (tools/kbench.py, one window):
# instructions, VALU-active / wait / stall quad-cycles -- what the VALU-issue floor in bench.py's roofline.floors is made of.
#   bash tools/collect_pmc_bench_mix.sh <tag>  ->  gpurun_out/<tag>/pmc_mix/ ; python tools/summarize_pmc_bench_mix.py gpurun_out/<tag>/pmc_mix r04
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG/pmc_mix
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps ${MIX_STEPS:-3} --warmup 1 --no-supervisor --extra none --no-cpu-baseline"
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
pass() {  # name "counters" command...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$O/$name.d" -o p -- "$@" > "$O/$name.json" 2> "$O/$name.err"
  f=$(find "$O/$name.d" -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then reduce "$f" "$O/$name.csv"; else echo "no counter file for $name" >> "$O/errors.log"; tail -5 "$O/$name.err" >> "$O/errors.log"; fi
  rm -rf "$O/$name.d"
}
pass mix.FETCH_SIZE FETCH_SIZE $BENCH
pass mix.WRITE_SIZE WRITE_SIZE $BENCH
pass mix.SQ_A "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES" $BENCH
pass mix.SQ_B "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" $BENCH
DNE_NSUB=1 pass alone.SQ_A "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES" python "$R/tools/kbench.py" --pairs 2500 --reps 1 --tslimit 6
DNE_NSUB=1 pass alone.SQ_B "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" python "$R/tools/kbench.py" --pairs 2500 --reps 1 --tslimit 6
ls -la "$O"; cat "$O/errors.log" 2>/dev/null
