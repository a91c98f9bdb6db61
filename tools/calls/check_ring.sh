#!/bin/bash
# staged: stop at the first failure (a faulting kernel must not burn the remaining passes' timeouts)
TAG=${1:-r05g}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
CLK=$R/deep-neuroevolution_amd/csrc/libdne_hip_clock.so
cd $R
timeout 120 python -m pytest tests/test_gpu_edges.py -x -q -k "test_every_step_kernel_variant and (knobs2- or knobs20 or knobs21 or knobs22 or knobs23 or knobs24 or knobs25)" > $O/pytest_ring.log 2>&1 || { echo "ring subset FAILED"; tail -15 $O/pytest_ring.log | cut -c1-300; exit 1; }
tail -2 $O/pytest_ring.log
timeout 300 python -m pytest tests/test_gpu_edges.py -x -q -k "test_reference_pass_under or test_every_step_kernel_variant" > $O/pytest_edges.log 2>&1 || { echo "edges FAILED"; tail -15 $O/pytest_edges.log | cut -c1-300; exit 1; }
tail -2 $O/pytest_edges.log
timeout 300 python -m pytest tests/test_gpu_fullsize.py -x -q -k "test_full_generation_bit_exact" > $O/pytest_full_ring.log 2>&1 || { echo "full FAILED"; tail -15 $O/pytest_full_ring.log | cut -c1-300; exit 1; }
tail -2 $O/pytest_full_ring.log
timeout 300 python tools/ab_inproc.py $AB_SETTINGS --rounds 2 > $O/ab.jsonl 2> $O/ab.err || { echo "ab FAILED"; tail -5 $O/ab.err; exit 1; }
tail -1 $O/ab.jsonl
env DNE_LIB_PATH=$CLK DNE_NSUB=1 timeout 120 python tools/duo_tick_clock.py > "$O/tick.ring.json" 2> "$O/tick.ring.err" || { echo "tick FAILED"; exit 1; }
head -c 1100 $O/tick.ring.json
