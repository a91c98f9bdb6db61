#!/bin/bash
TAG=${1:-r04i}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
for cfg in "X=0" "DNE_CONV_FUSED_MIN=65" "DNE_CONV_FUSED_MIN=65 DNE_FC_SUB_NSUB=3"; do
  env $cfg timeout 300 python tools/ga_lockstep_profile.py > "$O/ga_prof.$cfg.json" 2> "$O/ga_prof.$cfg.err"
  python - "$O/ga_prof.$cfg.json" "$cfg" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); g=d["generation_1"]
print(sys.argv[2], "gen1 %.1f ms %.0f steps/s" % (g["wall_ms"], g["steps_per_s"]), d["lock_step_us_at_width"])
PY
done
timeout 600 python tools/ab_inproc.py --pairs 312 --skip alone --gens 10 "X=0" "DNE_CONV_FUSED_MIN=65" "DNE_CONV_FUSED_MIN=33 DNE_CONV12T_MAX=32" > $O/ab_312.jsonl 2> $O/ab_312.err; tail -1 $O/ab_312.jsonl
timeout 600 python tools/ab_inproc.py --skip alone,lockstep --gens 8 "X=0" "DNE_CONV_FUSED_MIN=65" > $O/ab_2500.jsonl 2> $O/ab_2500.err; tail -1 $O/ab_2500.jsonl
