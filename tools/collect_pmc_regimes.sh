#!/bin/bash
# HBM-side traffic of the streaming fc kernel (k_fc_duo) at the window shapes the bench really runs -- FETCH_SIZE and WRITE_SIZE, each
# in its own rocprofv3 --pmc pass (kernel trace only), over a workload whose every lock-step has the same width (tools/kbench.py:
# nobody dies within 6 steps).  Counter collection serialises the dispatches: a launch is measured with the chip to itself.
#   regime               pairs  windows   units per launch (member-steps)
#   full_1window          2500     1       5000    k_fc_ring (since round 5): one unit per wave, eight per workgroup
#   full_3windows         2500     3       1667    round 2's shape of the first lock-steps
#   full_4windows         2500     4       1250    the bench's first lock-steps (>= 1900 active pairs) since the sweep
#   half_4windows         1250     4        625    one unit per wave (below 1500 active pairs)
#   third_4windows         800     4        400    the lower end of the streaming regime
# tools/summarize_pmc_regimes.py gpurun_out/<tag>/pmc_regimes rNN  ->  profiles/rNN_pmc.json
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG/pmc_regimes
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run() {  # name pairs nsub
  for c in FETCH_SIZE WRITE_SIZE; do
    DNE_NSUB=$3 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/$1.$c" -o kb -- python "$R/tools/kbench.py" --pairs $2 --reps 1 --tslimit 6 > "$O/$1.$c.log" 2>&1
    f=$(find "$O/$1.$c" -name '*counter_collection.csv' | head -1)
    python - "$f" "$O/$1.$c.csv" <<'PY'
import csv, collections, sys
by = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = (r['Kernel_Name'].split('(')[0].replace('void ', ''), int(r.get('Grid_Size', r.get('Grid_Size_X', 0))))
    by[k][r['Dispatch_Id']] += float(r['Counter_Value'])
with open(sys.argv[2], 'w') as f:
    f.write("kernel,grid_size,dispatches,avg_counter_KB\n")
    for k, v in sorted(by.items()):
        f.write('"%s",%d,%d,%.1f\n' % (k[0], k[1], len(v), sum(v.values()) / len(v)))
PY
    rm -rf "$O/$1.$c"
  done
}
# REGIMES="full_1window full_4windows" restricts the collection (a pass costs ~25 s of box time)
for spec in "full_1window 2500 1" "full_3windows 2500 3" "full_4windows 2500 4" "half_4windows 1250 4" "third_4windows 800 4"; do
  set -- $spec
  if [ -z "${REGIMES:-}" ] || echo " $REGIMES " | grep -q " $1 "; then run $1 $2 $3; fi
done
ls "$O"
