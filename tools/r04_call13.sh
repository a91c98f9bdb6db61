#!/bin/bash
TAG=${1:-r04m}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_edges.py -x -q -k "variant and not ga" > $O/pytest_variants.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_variants.log
timeout 600 python tools/ab_inproc.py --gens 8 --rounds 2 "X=0" "DNE_DUO_FAT=1" "DNE_DUO_FAT=1 DNE_NSUB_FULL=3" "DNE_DUO_FAT=1 DNE_FC_GRID=256" "DNE_DUO_FAT=1 DNE_FC_PRIO=0" > $O/ab.jsonl 2> $O/ab.err; echo "ab rc=$?"; cat $O/ab.jsonl | tail -1; tail -2 $O/ab.err
