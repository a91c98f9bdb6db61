#!/bin/bash
TAG=${1:-r05p}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 600 python tools/ab_inproc.py "X=0" "DNE_OUT_LDS_KB=32" "DNE_OUT_LDS_KB=64" "DNE_RENDER_THREADS=640" --rounds 2 --gens 6 --skip alone,lockstep > $O/ab.jsonl 2> $O/ab.err || { echo "ab FAILED"; tail -5 $O/ab.err; exit 1; }
tail -1 $O/ab.jsonl
