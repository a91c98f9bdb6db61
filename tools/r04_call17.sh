#!/bin/bash
TAG=${1:-r04r}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
timeout 900 python tools/ab_inproc.py --skip alone --gens 8 --rounds 2 "X=0" "DNE_DUO_HEAD_FUSED=1" "DNE_RENDER_THREADS=512" "DNE_CONV_FUSED=0" > $O/ab.jsonl 2> $O/ab.err; tail -1 $O/ab.jsonl
