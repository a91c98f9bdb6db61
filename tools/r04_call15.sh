#!/bin/bash
TAG=${1:-r04o}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
timeout 600 python tools/ab_inproc.py --pairs 312 --skip alone --gens 10 --rounds 2 "X=0" "DNE_FC_DUO_MIN=250" "DNE_FC_DUO_MIN=150" > $O/ab_312.jsonl 2> $O/ab_312.err; tail -1 $O/ab_312.jsonl
timeout 600 python tools/ab_inproc.py --pairs 625 --skip alone --gens 10 --rounds 1 "X=0" "DNE_FC_DUO_MIN=300" > $O/ab_625.jsonl 2> $O/ab_625.err; tail -1 $O/ab_625.jsonl
