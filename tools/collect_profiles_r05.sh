#!/bin/bash
# Round 5's collection (a trimmed tools/collect_profiles.sh: the box budget went into kernel work): everything profiles/r05_* and DESIGN.md
# section 9 quote, from ONE run on the final tree.   gpurun -- 'bash tools/collect_profiles_r05.sh r05z'
#   1. counters first, summarised on the box into profiles/r05_pmc.json so that the bench lines quote the file that gets committed:
#      FETCH_SIZE / WRITE_SIZE per fixed-width regime (tools/collect_pmc_regimes.sh) and on bench.py's own launch mix + the SQ groups
#      (tools/collect_pmc_bench_mix.sh) for the roofline kernel k_fc_ring
#   2. the driver's command, the default command, rocprofv3 --kernel-trace --stats of the default command
#   3. the vector-memory path's counters of k_fc_ring alone (tools/collect_pmc_duo_mem.sh, TA / TCP / TCC)
#   4. population shares (what a rank sees at N = 2 / 4 / 8)
#   5. counted traffic of the GA legs (tools/ga_bench.py, --large)
#   6. emulator-cost sensitivity (DNE_ENV_BURN, profiling build)
set -u
TAG=${1:-r05z}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
PFX=r05
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
REGIMES="full_1window full_4windows" bash "$R/tools/collect_pmc_regimes.sh" "$TAG" > "$O/pmc_regimes.log" 2>&1
bash "$R/tools/collect_pmc_bench_mix.sh" "$TAG" > "$O/pmc_mix.log" 2>&1
( cd "$R" && python tools/summarize_pmc_regimes.py "$O/pmc_regimes" "$PFX" > "$O/pmc_summary.log" 2>&1 && python tools/summarize_pmc_bench_mix.py "$O/pmc_mix" "$PFX" >> "$O/pmc_summary.log" 2>&1 && cp "profiles/${PFX}_pmc.json" "$O/${PFX}_pmc.json" )
cat "$O/pmc_summary.log"
# the GA legs' counters before the bench lines too (the line's extra.ga / ga_large quote profiles/r05_pmc_ga.json)
mkdir -p "$O/pmc_ga"
reduce() {
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
for leg in ga ga_large; do
  flag=""; [ $leg = ga_large ] && flag="--large"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_ga/$leg.$c.d" -o p -- python "$R/tools/ga_bench.py" $flag > "$O/pmc_ga/$leg.$c.json" 2> "$O/pmc_ga/$leg.$c.err"
    f=$(find "$O/pmc_ga/$leg.$c.d" -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && reduce "$f" "$O/pmc_ga/$leg.$c.csv"
    rm -rf "$O/pmc_ga/$leg.$c.d"
  done
done
( cd "$R" && python tools/summarize_pmc_ga.py "$O/pmc_ga" "$PFX" > "$O/pmc_ga_summary.log" 2>&1; cp "profiles/${PFX}_pmc_ga.json" "$O/" 2>/dev/null; cat "$O/pmc_ga_summary.log" )
cd /tmp
python "$R/bench.py" --steps 20 --warmup 5 > "$O/bench_driver_cmd.json" 2> "$O/bench_driver_cmd.err"
python "$R/bench.py" --extra none > "$O/bench_default.json" 2> "$O/bench_default.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o bench -- python "$R/bench.py" --no-cpu-baseline --no-supervisor --extra none > "$O/bench_profiled.json" 2> "$O/prof.err"
cp "$(find "$O/stats" -name '*kernel_stats.csv' | head -1)" "$O/bench_kernel_stats.csv" 2>/dev/null
find "$O/stats" -name "*kernel_trace.csv" -delete
MIX=0 FULL=0 SKIP="TAb SQa SQb TCCc" bash "$R/tools/collect_pmc_duo_mem.sh" "$TAG" > "$O/pmc_ring_mem.log" 2>&1
( cd "$R" && python tools/summarize_pmc_duo_mem.py "$O/pmc_duo_mem" "$PFX" > "$O/pmc_ring_mem_summary.txt" 2>&1; cp "profiles/${PFX}_pmc_fc_ring_mem.json" "$O/" 2>/dev/null )
for p in 2500 1250 624; do python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --extra none --pop $p 2>/dev/null | tail -1; done > "$O/population_shares.jsonl"
CLK=$R/deep-neuroevolution_amd/csrc/libdne_hip_clock.so
for b in 0 2000 8000 20000; do
  DNE_LIB_PATH=$CLK DNE_ENV_BURN=$b python "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-supervisor --extra none 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(json.dumps({'lane_instructions_per_raw_frame': $b, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'env_ms_per_generation': d['stage_ms_per_generation'].get('env_ms')}))"
done > "$O/env_burn.jsonl"
cat "$O/env_burn.jsonl"
find "$O" -name "*.csv" -size +20M -delete
ls "$O"; tail -c 300 "$O/bench_driver_cmd.json"
