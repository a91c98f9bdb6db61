#!/bin/bash
TAG=${1:-r04t}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_edges.py -x -q -k "variant and not ga" > $O/pytest_variants.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_variants.log
timeout 900 python tools/ab_inproc.py --skip alone,lockstep --gens 10 --rounds 2 "X=0" "DNE_BURST_TAIL=32" "DNE_BURST_TAIL=64" "DNE_BURST=8" "DNE_BURST=32" "DNE_BURST=8 DNE_BURST_TAIL=32" > $O/ab.jsonl 2> $O/ab.err; tail -1 $O/ab.jsonl
