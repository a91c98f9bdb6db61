#!/bin/bash
TAG=${1:-r05j}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 600 python tools/ab_inproc.py "X=0" "DNE_NSUB_FULL=3" "DNE_FC_PRIO=0" "DNE_FC_PRIO=1" "DNE_DUO_SOLO_BELOW=1100" "DNE_DUO_SOLO_BELOW=1900" "DNE_NSUB_MID=3" --rounds 2 --gens 6 --skip alone > $O/ab.jsonl 2> $O/ab.err || { echo "ab FAILED"; tail -5 $O/ab.err; exit 1; }
tail -1 $O/ab.jsonl
