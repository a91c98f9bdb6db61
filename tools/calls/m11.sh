#!/bin/bash
TAG=${1:-r05o}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 600 python tools/ab_inproc.py "X=0" "DNE_RENDER_THREADS=512" "DNE_RENDER_THREADS=768" "DNE_RENDER_THREADS=1024" "DNE_RENDER_THREADS=512 DNE_NSUB_FULL=3" --rounds 2 --gens 6 --skip alone,lockstep > $O/ab.jsonl 2> $O/ab.err || { echo "ab FAILED"; tail -5 $O/ab.err; exit 1; }
tail -1 $O/ab.jsonl
