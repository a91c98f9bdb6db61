#!/bin/bash
# Round 6's collection on the FINAL tree (one run): everything profiles/r06_* of the bench line and DESIGN.md section 9 quote.
#   gpurun -- 'bash tools/collect_profiles_r06.sh r06z'
#   1. the GPU suite + smoke (stop at a failure)
#   2. counters first, summarised on the box into profiles/r06_pmc.json / r06_pmc_ga.json so that the bench lines quote the files that get committed:
#      FETCH_SIZE / WRITE_SIZE per fixed-width regime, on bench.py's own launch mix + the SQ groups of k_fc_ring; the GA legs' counted traffic
#   3. the driver's command (with extra.predicted_n2/4/8, extra.shares), the default command, rocprofv3 --kernel-trace --stats of the default command
set -u
TAG=${1:-r06z}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
PFX=r06
mkdir -p "$O"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
[ $rc -ne 0 ] && exit 1
DNE_TEST_VARIANTS=1 timeout 900 python -m pytest tests -m "gpu and variants" -x -q > $O/pytest_variants.log 2>&1; echo "variants rc=$?"; grep -E "passed|failed" $O/pytest_variants.log | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
REGIMES="full_1window full_4windows" bash "$R/tools/collect_pmc_regimes.sh" "$TAG" > "$O/pmc_regimes.log" 2>&1
bash "$R/tools/collect_pmc_bench_mix.sh" "$TAG" > "$O/pmc_mix.log" 2>&1
( cd "$R" && python tools/summarize_pmc_regimes.py "$O/pmc_regimes" "$PFX" > "$O/pmc_summary.log" 2>&1 && python tools/summarize_pmc_bench_mix.py "$O/pmc_mix" "$PFX" >> "$O/pmc_summary.log" 2>&1 && cp "profiles/${PFX}_pmc.json" "$O/${PFX}_pmc.json" )
cat "$O/pmc_summary.log"
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
cp "$(find "$O/stats" -name '*kernel_stats.csv' | head -1)" "$O/bench_kernel_stats.csv" 2>/dev/null; rm -rf "$O/stats"
DNE_NSUB=1 python "$R/tools/kbench.py" --pairs 2500 --reps 2 --tslimit 12 > "$O/kbench_alone.json" 2>&1
python - <<PY
import json
def last(p): return json.loads([l for l in open(p) if l.startswith("{")][-1])
d=last("$O/bench_driver_cmd.json"); r=d["roofline"]; print("driver:", d["value"], d["ms_per_step"], r["avg_launch_ms"], "frac", r["frac"], "frac_counter", r["frac_counter"], "alg", r["frac_algorithmic"])
print("shares", {k: v["ms_per_generation"] for k, v in d["extra"]["shares"].items()}, "predicted", {k: round(v["value"]) for k, v in d["extra"].items() if k.startswith("predicted")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["gpu_over_cpu"])
d=last("$O/bench_default.json"); print("default:", d["value"], d["ms_per_step"])
PY
head -8 $O/bench_kernel_stats.csv | cut -c1-150
