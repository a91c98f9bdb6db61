#!/bin/bash
TAG=${1:-r04g}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
timeout 600 python tools/ab_inproc.py --skip alone --gens 8 "X=0" "DNE_DUO_LDS_KB=84" "DNE_DUO_LDS_KB=84 DNE_NSUB_FULL=3" "DNE_DUO_LDS_KB=50" > $O/ab_lds.jsonl 2> $O/ab_lds.err; tail -1 $O/ab_lds.jsonl
timeout 600 python tools/ab_inproc.py --pairs 312 --skip alone --gens 10 "X=0" "DNE_FC_SUB=2 DNE_FC_SUB_PRIO=3 DNE_FC_SUB_GRID=100000 DNE_FC_SUB_HEAD=0" "DNE_FC_SUB=2 DNE_FC_SUB_PRIO=3 DNE_FC_SUB_GRID=100000 DNE_FC_SUB_HEAD=0 DNE_FC_SUB_NSUB=3" "DNE_FC_SUB=2 DNE_FC_SUB_GRID=100000 DNE_FC_SUB_NSUB=3" > $O/ab_312.jsonl 2> $O/ab_312.err; tail -1 $O/ab_312.jsonl
timeout 600 python tools/ab_inproc.py --pairs 625 --skip alone --gens 10 "X=0" "DNE_FC_SUB=2 DNE_FC_SUB_PRIO=3 DNE_FC_SUB_GRID=100000 DNE_FC_SUB_HEAD=0 DNE_FC_SUB_NSUB=3" "DNE_FC_SUB=2 DNE_FC_SUB_GRID=100000 DNE_FC_SUB_NSUB=3" > $O/ab_625.jsonl 2> $O/ab_625.err; tail -1 $O/ab_625.jsonl
