#!/bin/bash
# round 4, third GPU call: k_fc_sub (sub-slice fc) bit-exactness + GA / ES mid-range A/B; XCD-aware work-item mapping of k_fc_duo
TAG=${1:-r04c}
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_edges.py -x -q -k "variant" > $O/pytest_variants.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_variants.log
for cfg in "DNE_FC_SUB=0" "X=0" "DNE_FC_SUB_NSUB=1" "DNE_FC_SUB_NSUB=3" "DNE_FC_SUB_MIN=49"; do
  env $cfg timeout 300 python tools/ga_lockstep_profile.py > $O/ga_prof.$cfg.json 2> $O/ga_prof.$cfg.err
  python - "$O/ga_prof.$cfg.json" "$cfg" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); g=d["generation_1"]
print(sys.argv[2], "gen1 %.1f ms %.0f steps/s" % (g["wall_ms"], g["steps_per_s"]), d["lock_step_us_at_width"])
PY
done
timeout 600 python tools/ab_inproc.py --pairs 312 --skip alone --gens 10 "X=0" "DNE_FC_SUB=2" "DNE_FC_SUB=2 DNE_FC_SUB_NSUB=3" > $O/ab_312.jsonl 2> $O/ab_312.err; tail -1 $O/ab_312.jsonl
timeout 600 python tools/ab_inproc.py --pairs 625 --skip alone --gens 10 "X=0" "DNE_FC_SUB=2" "DNE_FC_SUB=2 DNE_FC_SUB_NSUB=3" > $O/ab_625.jsonl 2> $O/ab_625.err; tail -1 $O/ab_625.jsonl
timeout 600 python tools/ab_inproc.py "X=0" "DNE_DUO_XCD=1" "DNE_FC_SUB=2" > $O/ab_xcd.jsonl 2> $O/ab_xcd.err; tail -1 $O/ab_xcd.jsonl
