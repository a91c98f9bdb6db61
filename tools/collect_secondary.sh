#!/bin/bash
# Secondary evidence for profiles/: GA and NS-ES evaluation benches (+ kernel stats of the GA one) and the per-rank share of
# the headline population at N = 2 / 4 / 8 (what one rank evaluates; the real multi-GPU runs are the driver's).
set -u
TAG=${1:-r01s}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python "$R/tools/ga_bench.py" > "$O/ga_bench.jsonl" 2> "$O/ga_bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ga_stats" -o ga -- python "$R/tools/ga_bench.py" > /dev/null 2> "$O/ga_prof.err"
python "$R/tools/nses_bench.py" > "$O/nses_bench.jsonl" 2> "$O/nses_bench.err"
for p in 2500 1250 624; do python "$R/bench.py" --no-cpu-baseline --pop $p 2>/dev/null | tail -1; done > "$O/population_shares.jsonl"
python "$R/tools/len_profile.py" --pairs 312 > "$O/len_profile_312.json" 2>/dev/null
python "$R/tools/len_profile.py" --pairs 2500 > "$O/len_profile_2500.json" 2>/dev/null
find "$O" -name "*.csv" -size +20M -delete
find "$O" -name "*kernel_trace.csv" -delete
ls -la "$O" "$O/ga_stats" | head -30
