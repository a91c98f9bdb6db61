#!/bin/bash
TAG=${1:-r04l}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
timeout 600 python tools/ab_inproc.py --skip lockstep,gen --rounds 2 "X=0" "DNE_FC_GRID=256" "DNE_FC_GRID=384" "DNE_FC_GRID=128" > $O/ab.jsonl 2> $O/ab.err; cat $O/ab.jsonl | tail -1
