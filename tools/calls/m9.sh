#!/bin/bash
TAG=${1:-r05m}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 600 python tools/ab_inproc.py "X=0" "DNE_DUO_HEAD_FUSED=1" "DNE_DUO_GRID=256" "DNE_BURST=48" "DNE_BURST=64" "DNE_FC_DUO_MIN=800" --rounds 2 --gens 6 --skip alone,lockstep > $O/ab.jsonl 2> $O/ab.err || { echo "ab FAILED"; tail -5 $O/ab.err; exit 1; }
tail -1 $O/ab.jsonl
