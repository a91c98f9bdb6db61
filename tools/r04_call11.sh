#!/bin/bash
TAG=${1:-r04k}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
timeout 600 python tools/ab_inproc.py --skip alone --gens 8 --rounds 1 "X=0" "DNE_DUO_W=4 DNE_DUO_GRID=512" "DNE_DUO_W=4 DNE_DUO_GRID=640" "DNE_DUO_W=4 DNE_DUO_GRID=512 DNE_NSUB_FULL=3" "DNE_FC_GRID=384" > $O/ab.jsonl 2> $O/ab.err; cat $O/ab.jsonl | tail -1
