#!/bin/bash
TAG=${1:-r05n}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 600 python tools/ab_inproc.py "X=0" "DNE_CONV_FUSED=0" "DNE_RENDER_THREADS=512" --rounds 2 --gens 6 --skip alone,lockstep > $O/ab.jsonl 2> $O/ab.err || { echo "ab FAILED"; tail -5 $O/ab.err; exit 1; }
tail -1 $O/ab.jsonl
timeout 300 python bench.py --steps 20 --warmup 5 --extra none --no-cpu-baseline > $O/bench_driver_noextra.json 2> $O/bench.err; tail -c 600 $O/bench_driver_noextra.json | head -c 300; python - <<PY
import json
d=json.loads([l for l in open("$O/bench_driver_noextra.json") if l.startswith("{")][-1]); print("driver cmd (no extras):", d["value"], d["ms_per_step"])
PY
