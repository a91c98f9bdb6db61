#!/bin/bash
# Round 6: (1) the driver's command with the new extras (predicted_n2/4/8 + shares, the longer CPU sample) -- how long does the whole line take;
# (2) k_conv12 at 126 registers (tools/experiments/conv12_lean.patch, csrc/ab/libdne_hip_lean.so) under 2 / 3 / 4 windows: does co-residency
#     with k_fc_ring pay under ANY window count (r06b: slower at four)
TAG=${1:-r06d}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
( time python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err ) 2> $O/bench_full.time; tail -3 $O/bench_full.time
python - <<PY
import json
d = json.loads(open("$O/bench_full.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms", round(d["ms_per_step"], 2), "frac", d["roofline"]["frac"], d["roofline"].get("frac_basis"), "alg", d["roofline"]["frac_algorithmic"])
c = d["cpu_baseline"]; print("cpu", round(c["value"]), c["cores"], c["sample"][:200]); print([ (r["workers"], r["items"], r["wall_s"], round(r["rate_wall"])) for r in c["sweep"]])
ex = d["extra"]
for k, v in ex.items():
    if k == "shares": print("shares", v); continue
    print(k, v.get("error") or (round(v.get("value", 0)), v.get("ms_per_step"), v.get("theta_matches_one_rank_evaluation"), v.get("bench_wall_s")))
PY
grep -E "^\[bench" $O/bench_full.err | tail -25 | cut -c1-150
for lib in product lean; do
  if [ $lib = lean ]; then export DNE_LIB_PATH=$R/deep-neuroevolution_amd/csrc/ab/libdne_hip_lean.so; else unset DNE_LIB_PATH; fi
  timeout 600 python tools/ab_inproc.py "X=0" "DNE_NSUB_FULL=2" "DNE_NSUB_FULL=3" "DNE_NSUB_FULL=2 DNE_FC_PRIO=0" --rounds 2 --gens 6 --skip alone > $O/ab_$lib.jsonl 2> $O/ab_$lib.err
  tail -1 $O/ab_$lib.jsonl | python -c "
import json,sys
for k,v in json.loads(sys.stdin.read())['summary'].items(): print('$lib', k, v['lockstep_ms'], v['gen_ms'], v['theta_sha'])"
done
