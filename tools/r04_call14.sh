#!/bin/bash
TAG=${1:-r04n}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
timeout 600 python tools/ab_inproc.py --skip alone,lockstep --gens 8 --rounds 2 "X=0" "DNE_DUO_SOLO_BELOW=1000" "DNE_DUO_SOLO_BELOW=2000" "DNE_NSUB_MID=3" "DNE_FC_DUO_MIN=500" "DNE_DUO_FAT=0" > $O/ab_2500.jsonl 2> $O/ab_2500.err; tail -1 $O/ab_2500.jsonl
timeout 600 python tools/ab_inproc.py --pairs 625 --skip alone --gens 10 --rounds 2 "X=0" "DNE_FC_DUO_MIN=450" "DNE_FC_DUO_MIN=450 DNE_NSUB_MID=3" "DNE_FC_DUO_MIN=450 DNE_NSUB_MID=2" > $O/ab_625.jsonl 2> $O/ab_625.err; tail -1 $O/ab_625.jsonl
timeout 600 python tools/ab_inproc.py --pairs 1250 --skip alone --gens 10 --rounds 2 "X=0" "DNE_DUO_FAT=0" "DNE_NSUB_MID=3" > $O/ab_1250.jsonl 2> $O/ab_1250.err; tail -1 $O/ab_1250.jsonl
timeout 300 python tools/ab_inproc.py --skip lockstep,gen --rounds 1 --idx-align 4 "X=0" "DNE_DUO_FAT=0" > $O/ab_align4.jsonl 2> $O/ab_align4.err; tail -1 $O/ab_align4.jsonl
timeout 300 python tools/ab_inproc.py --skip lockstep,gen --rounds 1 --idx-align 64 "X=0" > $O/ab_align64.jsonl 2> $O/ab_align64.err; tail -1 $O/ab_align64.jsonl
