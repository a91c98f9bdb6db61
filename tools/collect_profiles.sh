#!/bin/bash
# Run on the MI355X box (gpurun -- 'bash tools/collect_profiles.sh r02p'): everything profiles/ and DESIGN.md sections 4 / 9 quote.
#   bench line (defaults) and the driver's command (--steps 20 --warmup 5); rocprofv3 kernel stats of the default command;
#   the two HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE, each in its own run, kernel-trace only) over one full-width lock-step
#   workload; the matrix-core PMC pass (tools/collect_pmc_mfma.sh); the env-free micro-benchmarks; GA (small network and the GPU tree's
#   LargeModel, with its kernel stats) / NS-ES benches; the six-game sweep;
#   population shares; lock-step length profiles; tail latency.
# Outputs land in gpurun_out/<tag>/; `python tools/refresh_profiles.py gpurun_out/<tag> r02` copies the summaries into profiles/.
set -u
TAG=${1:-r02p}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python "$R/bench.py" > "$O/bench_default.json" 2> "$O/bench_default.err"
python "$R/bench.py" --steps 20 --warmup 5 > "$O/bench_driver_cmd.json" 2> "$O/bench_driver_cmd.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o bench -- python "$R/bench.py" --no-cpu-baseline --no-supervisor > "$O/bench_profiled.json" 2> "$O/prof.err"
# traffic of the streaming fc kernel: one window (DNE_NSUB=1) so that a launch covers exactly 2500 pairs = 5000 member-steps
DNE_NSUB=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch" -o kb -- python "$R/tools/kbench.py" --reps 1 --tslimit 6 > "$O/pmc_fetch.log" 2>&1
DNE_NSUB=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write" -o kb -- python "$R/tools/kbench.py" --reps 1 --tslimit 6 > "$O/pmc_write.log" 2>&1
bash "$R/tools/collect_pmc_mfma.sh" "$TAG" > /dev/null 2>&1
python "$R/tools/micro_bench.py" > "$O/micro.json" 2> "$O/micro.err"
python "$R/tools/ga_bench.py" > "$O/ga_bench.jsonl" 2> "$O/ga_bench.err"
python "$R/tools/nses_bench.py" > "$O/nses_bench.jsonl" 2> "$O/nses_bench.err"
python "$R/tools/ga_bench.py" --large > "$O/ga_large_bench.jsonl" 2> "$O/ga_large_bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ga_large_stats" -o g -- python "$R/tools/ga_bench.py" --large > /dev/null 2>&1
cp "$(find "$O/ga_large_stats" -name '*kernel_stats.csv' | head -1)" "$O/ga_large_kernel_stats.csv" 2>/dev/null
python "$R/tools/six_game_sweep.py" > "$O/six_game_sweep.jsonl" 2> "$O/six_game_sweep.err"
for p in 2500 1250 624; do python "$R/bench.py" --no-cpu-baseline --pop $p 2>/dev/null | tail -1; done > "$O/population_shares.jsonl"
python "$R/tools/len_profile.py" --pairs 312 > "$O/len_profile_312.json" 2>/dev/null
python "$R/tools/len_profile.py" --pairs 2500 > "$O/len_profile_2500.json" 2>/dev/null
python "$R/tools/tail_bench.py" > "$O/tail_bench.json" 2>/dev/null
find "$O" -name "*.csv" -size +20M -delete    # traces can be large; the summaries are what travels back
find "$O" -name "*kernel_trace.csv" -size +2M -delete
ls "$O"; tail -c 400 "$O/bench_driver_cmd.json"
