#!/bin/bash
# Deep GA (config 3): the sub-slice regime's knobs (97 .. 320 children alive = 62 % of a generation's time): windows, grid, range.
#   SETTINGS="X=0|DNE_FC_SUB_NSUB=4|DNE_FC_SUB_NSUB=4 DNE_FC_SUB_MAX=500" bash tools/calls/r06_ga_knobs.sh <tag>
TAG=${1:-r06x}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
IFS='|' read -ra ARR <<< "${SETTINGS:-X=0|DNE_FC_SUB_NSUB=3|DNE_FC_SUB_NSUB=4|X=0}"
for s in "${ARR[@]}"; do
  v=$(env $s timeout 200 python tools/ga_bench.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']))")
  echo "$s $v" | tee -a $O/ga_knobs.txt
done
