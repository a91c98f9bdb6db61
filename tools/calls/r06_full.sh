#!/bin/bash
# Round 6: the whole GPU suite on the current tree, the driver's command, and the N = 2 / 4 / 8 rehearsal under the default gates
TAG=${1:-r06h}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
( time python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err ) 2> $O/bench_full.time; tail -3 $O/bench_full.time
python - <<PY
import json
d = json.loads(open("$O/bench_full.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms", round(d["ms_per_step"], 2), "frac", round(d["roofline"]["frac"], 3), "alg", round(d["roofline"]["frac_algorithmic"], 3))
c = d["cpu_baseline"]; print("cpu", round(c["value"]), c["cores"], [(r["workers"], r["items"], r["wall_s"], round(r["rate_wall"])) for r in c["sweep"]])
ex = d["extra"]
for k, v in ex.items():
    if k == "shares": print("shares", {a: b["ms_per_generation"] for a, b in v.items()}); continue
    print(k, v.get("error") or (round(v.get("value", 0)), v.get("ms_per_step"), v.get("theta_matches_one_rank_evaluation"), v.get("shard"), round(v.get("bench_wall_s", 0), 1)))
PY
timeout 900 python tools/shard_ab.py --worlds 2,4,8 --gens 6 --warmup 3 "uniform" "table" "table DNE_RING_MIN=0" > $O/shard_ab.jsonl 2> $O/shard_ab.err
python - <<PY
import json
for l in open("$O/shard_ab.jsonl"):
    d = json.loads(l); print(d["world"], "%-40s" % d["setting"], d["ms_per_generation"], d["rank0_ms"], d["theta_sha"])
PY
