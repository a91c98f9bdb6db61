#!/bin/bash
TAG=${1:-r04p}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
timeout 900 python tools/ab_inproc.py --skip alone --gens 8 --rounds 2 "X=0" "DNE_STREAMS=5 DNE_NSUB_FULL=5" "DNE_STREAMS=6 DNE_NSUB_FULL=6" "DNE_STREAMS=8 DNE_NSUB_FULL=8" "DNE_STREAMS=6 DNE_NSUB_FULL=6 DNE_NSUB_MID=6" "DNE_NSUB_FULL=3" > $O/ab.jsonl 2> $O/ab.err; tail -1 $O/ab.jsonl
