#!/bin/bash
# Round 6: table-affine sharding (every rank draws its noise indices from its own 1/N stretch of the table) so that the table-ordered ring
# kernel shares rows on a rank's share as it does at one GPU -- rehearsed on one GPU for N = 2 / 4 / 8
TAG=${1:-r06f}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1500 python tools/shard_ab.py --worlds 2,4,8 --gens 6 --warmup 3 \
  "uniform" "table" "table DNE_FC_RING=2 DNE_FC_DUO_MIN=97" "table DNE_FC_RING=2 DNE_FC_DUO_MIN=97 DNE_DUO_SOLO_BELOW={share60}" \
  "table DNE_FC_RING=2 DNE_FC_DUO_MIN=97 DNE_DUO_SOLO_BELOW={share60} DNE_NSUB_MID=2" "table DNE_FC_RING=2 DNE_FC_DUO_MIN=97 DNE_DUO_SOLO_BELOW={share60} DNE_NSUB_MID=3" \
  "table DNE_FC_RING=2 DNE_FC_DUO_MIN=97 DNE_DUO_SOLO_BELOW={share60} DNE_LIST_SORT=1" \
  > $O/shard_ab.jsonl 2> $O/shard_ab.err
cut -c1-400 $O/shard_ab.jsonl; tail -3 $O/shard_ab.err
