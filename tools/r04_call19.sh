#!/bin/bash
TAG=${1:-r04u}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
timeout 900 python tools/ab_inproc.py --skip alone,lockstep --gens 10 --rounds 2 "X=0" "DNE_BURST=24" "DNE_BURST=32" "DNE_BURST=48" "DNE_BURST=64" > $O/ab.jsonl 2> $O/ab.err; tail -1 $O/ab.jsonl
