#!/bin/bash
# final tree: the GPU suite, smoke, then the bench lines / kernel stats / shares again (render workgroups of 512 threads)
TAG=${1:-r05y}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $O/pytest_gpu.log | cut -c1-200
[ $rc -ne 0 ] && exit 1
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-200
cd /tmp; export TMPDIR=/tmp
python "$R/bench.py" --steps 20 --warmup 5 > "$O/bench_driver_cmd.json" 2> "$O/bench_driver_cmd.err"
python "$R/bench.py" --extra none > "$O/bench_default.json" 2> "$O/bench_default.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o bench -- python "$R/bench.py" --no-cpu-baseline --no-supervisor --extra none > "$O/bench_profiled.json" 2> "$O/prof.err"
cp "$(find "$O/stats" -name '*kernel_stats.csv' | head -1)" "$O/bench_kernel_stats.csv" 2>/dev/null; rm -rf "$O/stats"
for p in 2500 1250 624; do python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --extra none --pop $p 2>/dev/null | tail -1; done > "$O/population_shares.jsonl"
python - <<PY
import json
def last(p): return json.loads([l for l in open(p) if l.startswith("{")][-1])
d=last("$O/bench_driver_cmd.json"); print("driver:", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac_counter"])
d=last("$O/bench_default.json"); print("default:", d["value"], d["ms_per_step"])
for l in open("$O/population_shares.jsonl"):
    x=json.loads(l); print(x["value"], x["ms_per_step"])
PY
head -6 $O/bench_kernel_stats.csv | cut -c1-150
