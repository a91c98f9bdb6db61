#!/bin/bash
TAG=${1:-r04w}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
for cfg in "X=0" "DNE_FC_PAD=1" "DNE_FC_PAD=2" "DNE_FC_PAD=2 DNE_FC_GRID=256"; do
  env $cfg timeout 300 python tools/ga_lockstep_profile.py > "$O/ga_prof.$cfg.json" 2> "$O/ga_prof.$cfg.err"
  python - "$O/ga_prof.$cfg.json" "$cfg" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); g=d["generation_1"]
print(sys.argv[2], "gen1 %.1f ms %.0f steps/s" % (g["wall_ms"], g["steps_per_s"]), d["lock_step_us_at_width"])
PY
done
