#!/bin/bash
# 30 generations of the bench's configuration three times with k_fc_ring and once with k_fc_duo: the theta digests must all be equal
# (a race in the ring's LDS-DMA hand-over would show as a digest that differs from run to run; a kernel that is not bit-exact as one that
# differs from k_fc_duo's)
TAG=${1:-r05d2}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp
for run in ring1 ring2 ring3 duo; do
  env $([ $run = duo ] && echo DNE_FC_RING=0 || echo X=0) timeout 300 python $R/bench.py --steps 30 --warmup 0 --extra none --no-cpu-baseline --no-supervisor > $O/$run.json 2> $O/$run.err
  echo "$run $(grep -o 'theta sha256 [0-9a-f]*' $O/$run.err | tail -1) $(python -c "import json;d=json.loads([l for l in open('$O/$run.json') if l.startswith('{')][-1]);print(d['value'], d['ms_per_step'])")"
done | tee $O/digests.txt
