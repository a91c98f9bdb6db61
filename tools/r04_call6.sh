#!/bin/bash
# k_fc_sub on a bounded grid: GA width profile and ES mid-range A/B over grid size / wave priority / windows
TAG=${1:-r04f}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_edges.py -x -q -k "variant and (SUB or knobs1)" > $O/pytest_variants.log 2>&1; tail -2 $O/pytest_variants.log
timeout 900 python -m pytest tests/test_gpu_edges.py -x -q -k "variant" > $O/pytest_variants.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_variants.log
for cfg in "X=0" "DNE_FC_SUB_PRIO=3" "DNE_FC_SUB_GRID=512" "DNE_FC_SUB_GRID=2048" "DNE_FC_SUB_GRID=512 DNE_FC_SUB_PRIO=3"; do
  env $cfg timeout 300 python tools/ga_lockstep_profile.py > "$O/ga_prof.$cfg.json" 2> "$O/ga_prof.$cfg.err"
  python - "$O/ga_prof.$cfg.json" "$cfg" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); g=d["generation_1"]
print(sys.argv[2], "gen1 %.1f ms %.0f steps/s" % (g["wall_ms"], g["steps_per_s"]), d["lock_step_us_at_width"])
PY
done
timeout 600 python tools/ab_inproc.py --pairs 312 --skip alone --gens 10 "X=0" "DNE_FC_SUB=2" "DNE_FC_SUB=2 DNE_FC_SUB_PRIO=3" "DNE_FC_SUB=2 DNE_FC_SUB_PRIO=3 DNE_FC_SUB_NSUB=3" "DNE_FC_SUB=2 DNE_FC_SUB_PRIO=3 DNE_FC_SUB_GRID=512" > $O/ab_312.jsonl 2> $O/ab_312.err; tail -1 $O/ab_312.jsonl
timeout 600 python tools/ab_inproc.py --pairs 625 --skip alone --gens 10 "X=0" "DNE_FC_SUB=2" "DNE_FC_SUB=2 DNE_FC_SUB_PRIO=3" "DNE_FC_SUB=2 DNE_FC_SUB_PRIO=3 DNE_FC_SUB_NSUB=3" "DNE_FC_SUB=2 DNE_FC_SUB_PRIO=3 DNE_FC_SUB_GRID=512" > $O/ab_625.jsonl 2> $O/ab_625.err; tail -1 $O/ab_625.jsonl
