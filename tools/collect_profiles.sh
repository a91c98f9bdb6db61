#!/bin/bash
# Run on the MI355X box (gpurun -- 'bash tools/collect_profiles.sh r01f'): bench line, rocprofv3 kernel stats of the same
# command, and the two PMC passes (FETCH_SIZE / WRITE_SIZE, each in its own run, kernel-trace only) over one full-width
# lock-step workload.  Outputs land in gpurun_out/<tag>/; tools/refresh_profiles.py copies the summaries into profiles/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python "$R/bench.py" > "$O/bench_default.json" 2> "$O/bench_default.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o bench -- python "$R/bench.py" --no-cpu-baseline > "$O/bench_profiled.json" 2> "$O/prof.err"
# traffic of the streaming fc kernel: one window (DNE_NSUB=1) so that a launch covers exactly 2500 pairs = 5000 member-steps
DNE_NSUB=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch" -o kb -- python "$R/tools/kbench.py" --reps 1 --tslimit 6 > "$O/pmc_fetch.log" 2>&1
DNE_NSUB=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write" -o kb -- python "$R/tools/kbench.py" --reps 1 --tslimit 6 > "$O/pmc_write.log" 2>&1
find "$O" -name "*.csv" -size +20M -delete    # traces can be large; the summaries are what travels back
ls -la "$O" "$O/stats" "$O/pmc_fetch" 2>/dev/null | head -40
tail -c 600 "$O/bench_default.json"
