#!/bin/bash
# Round 6: table-affine sharding, second matrix -- which streaming kernel on a rank's dense share (k_fc_duo solo / duo, k_fc_sub, k_fc_ring)
TAG=${1:-r06g}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1500 python tools/shard_ab.py --worlds 2,4,8 --gens 6 --warmup 3 \
  "table" "table DNE_FC_RING=0 DNE_DUO_SOLO_BELOW={share60}" "table DNE_FC_RING=0 DNE_FC_DUO_MIN=97" "table DNE_FC_RING=0 DNE_FC_DUO_MIN=97 DNE_DUO_SOLO_BELOW={share60}" \
  "table DNE_FC_RING=2 DNE_DUO_SOLO_BELOW={share60}" "table DNE_FC_DUO_MIN=200" "table DNE_NSUB_MID=3" \
  > $O/shard_ab.jsonl 2> $O/shard_ab.err
python - <<PY
import json
for l in open("$O/shard_ab.jsonl"):
    d = json.loads(l); print(d["world"], "%-90s" % d["setting"], d["ms_per_generation"], d["rank0_ms"], d["theta_sha"])
PY
tail -3 $O/shard_ab.err
