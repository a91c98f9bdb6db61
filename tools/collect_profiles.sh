#!/bin/bash
# Run on the MI355X box (gpurun -- 'bash tools/collect_profiles.sh r03p'): everything profiles/ and DESIGN.md sections 4 / 9 quote.
#   the driver's command (--steps 20 --warmup 5, with the extra block: GA on both networks, NS-ES, the six-game loop, config 1, CPU
#   baselines) and the default command; rocprofv3 kernel stats of the default command; the HBM-traffic PMC passes per regime
#   (tools/collect_pmc_regimes.sh: FETCH_SIZE / WRITE_SIZE, each in its own run, kernel-trace only); the matrix-core PMC pass
#   (tools/collect_pmc_mfma.sh); the env-free micro-benchmarks; population shares (what a rank sees at N = 2 / 4 / 8); lock-step
#   length profiles; tail latency at fixed width + its per-kernel durations, launch timeline and in-kernel phase clock; the
#   reference pass alone (chunked and as one chunk under rocprofv3), its kernels' in-kernel phase clock, shader clock / power under it and
#   under the bench (tools/clock_watch.py); the launch floor; the Deep-GA lock-step profiles.
# Outputs land in gpurun_out/<tag>/; `python tools/refresh_profiles.py gpurun_out/<tag> r03` copies the summaries into profiles/.
set -u
TAG=${1:-r04p}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
# the counters FIRST, summarised on the box into profiles/<prefix>_pmc.json, so that the bench lines below quote exactly the file that gets committed
# (round 3's tracked bench profile predated its PMC file: VERDICT round 3, weak 2)
PFX=${PFX:-r04}
bash "$R/tools/collect_pmc_regimes.sh" "$TAG" > "$O/pmc_regimes.log" 2>&1
bash "$R/tools/collect_pmc_bench_mix.sh" "$TAG" > "$O/pmc_mix.log" 2>&1
( cd "$R" && python tools/summarize_pmc_regimes.py "$O/pmc_regimes" "$PFX" > "$O/pmc_summary.log" 2>&1 && python tools/summarize_pmc_bench_mix.py "$O/pmc_mix" "$PFX" >> "$O/pmc_summary.log" 2>&1 && cp "profiles/${PFX}_pmc.json" "$O/${PFX}_pmc.json" )
cat "$O/pmc_summary.log"
cd /tmp
python "$R/bench.py" --steps 20 --warmup 5 > "$O/bench_driver_cmd.json" 2> "$O/bench_driver_cmd.err"
python "$R/bench.py" --extra none > "$O/bench_default.json" 2> "$O/bench_default.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o bench -- python "$R/bench.py" --no-cpu-baseline --no-supervisor --extra none > "$O/bench_profiled.json" 2> "$O/prof.err"
bash "$R/tools/collect_pmc_mfma.sh" "$TAG" > /dev/null 2>&1
"$R/tools/micro/valu_rate" > "$O/valu_rate.jsonl" 2>/dev/null
python "$R/tools/micro_bench.py" > "$O/micro.json" 2> "$O/micro.err"
for p in 2500 1250 624; do python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --extra none --pop $p 2>/dev/null | tail -1; done > "$O/population_shares.jsonl"
python "$R/tools/len_profile.py" --pairs 312 > "$O/len_profile_312.json" 2>/dev/null
python "$R/tools/len_profile.py" --pairs 2500 > "$O/len_profile_2500.json" 2>/dev/null
python "$R/tools/tail_bench.py" > "$O/tail_bench.json" 2>/dev/null
bash "$R/tools/tail_stats.sh" "$TAG" 1 8 24 > "$O/tail_stats.log" 2>&1
rocprofv3 --kernel-trace --output-format csv -d "$O/tl8" -o t -- python "$R/tools/tail_bench.py" 8 --steps 208 > /dev/null 2>&1
python "$R/tools/tail_timeline.py" "$O/tl8" > "$O/tail_timeline_8.json" 2>/dev/null; rm -rf "$O/tl8"
for p in 8 24; do DNE_LIB_PATH=$R/deep-neuroevolution_amd/csrc/libdne_hip_clock.so python "$R/tools/phase_clock.py" $p > "$O/phase_clock_$p.json" 2>/dev/null; done
python "$R/tools/ref_bench.py" > "$O/ref_bench.json" 2>/dev/null
REF_CHUNK=5000 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ref_alone" -o r -- python "$R/tools/ref_bench.py" > "$O/ref_bench_one_chunk.json" 2>/dev/null
cp "$(find "$O/ref_alone" -name '*kernel_stats.csv' | head -1)" "$O/ref_pass_one_chunk_kernel_stats.csv" 2>/dev/null; rm -rf "$O/ref_alone"
DNE_LIB_PATH=$R/deep-neuroevolution_amd/csrc/libdne_hip_clock.so REF_CHUNK=5000 python "$R/tools/ref_phase_clock.py" > "$O/ref_phase_clock.json" 2>/dev/null
REF_REPS=100 REF_CHUNK=5000 python "$R/tools/clock_watch.py" "$O/clock_watch_ref.csv" -- python "$R/tools/ref_bench.py" > /dev/null 2>&1
python "$R/tools/clock_watch.py" "$O/clock_watch_bench.csv" -- python "$R/bench.py" --steps 12 --warmup 3 --no-cpu-baseline --extra none > /dev/null 2>&1
"$R/tools/micro/launch_floor" > "$O/launch_floor.jsonl" 2>/dev/null
python "$R/tools/ga_lockstep_profile.py" > "$O/ga_lockstep_profile.json" 2> "$O/ga_lockstep_profile.err"
python "$R/tools/ga_lockstep_profile.py" --large > "$O/ga_large_lockstep_profile.json" 2> "$O/ga_large_lockstep_profile.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ga_large_stats" -o g -- python "$R/tools/ga_bench.py" --large > /dev/null 2>&1
cp "$(find "$O/ga_large_stats" -name '*kernel_stats.csv' | head -1)" "$O/ga_large_kernel_stats.csv" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ga_stats" -o g -- python "$R/tools/ga_bench.py" > /dev/null 2>&1
cp "$(find "$O/ga_stats" -name '*kernel_stats.csv' | head -1)" "$O/ga_kernel_stats.csv" 2>/dev/null
find "$O" -name "*.csv" -size +20M -delete    # traces can be large; the summaries are what travels back
find "$O" -name "*kernel_trace.csv" -size +2M -delete
ls "$O"; tail -c 400 "$O/bench_driver_cmd.json"
