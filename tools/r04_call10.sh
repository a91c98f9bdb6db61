#!/bin/bash
TAG=${1:-r04j}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_edges.py -x -q -k "variant and not ga" > $O/pytest_variants.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_variants.log
timeout 600 python tools/ab_inproc.py --gens 6 --rounds 1 "X=0" "DNE_FC_RING=1" > $O/ab_ring.jsonl 2> $O/ab_ring.err; echo "ab rc=$?"; cat $O/ab_ring.jsonl; tail -3 $O/ab_ring.err
