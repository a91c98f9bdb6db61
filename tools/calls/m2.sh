#!/bin/bash
# round-5 call 2: k_fc_ring + reference-pass overlap -- parity first, then same-box A/B, then the ring's tick clock
TAG=${1:-r05b}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
CLK=$R/deep-neuroevolution_amd/csrc/libdne_hip_clock.so
cd $R
timeout 600 python -m pytest tests/test_gpu_edges.py -x -q -k "test_reference_pass_under or test_every_step_kernel_variant" > $O/pytest_edges.log 2>&1; echo "edges rc=$?"; tail -5 $O/pytest_edges.log
DNE_FC_RING=1 timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "test_full_generation_bit_exact" > $O/pytest_full_ring.log 2>&1; echo "full ring rc=$?"; tail -3 $O/pytest_full_ring.log
timeout 600 python tools/ab_inproc.py "X=0" "DNE_FC_RING=1" "DNE_FC_RING=2" "DNE_FC_RING=2 DNE_DUO_FAT=0" "DNE_FC_RING=2 DNE_REF_OVERLAP=1" --rounds 2 > $O/ab.jsonl 2> $O/ab.err; echo "ab rc=$?"; tail -1 $O/ab.jsonl
env DNE_FC_RING=1 DNE_LIB_PATH=$CLK DNE_NSUB=1 timeout 200 python tools/duo_tick_clock.py > "$O/tick.ring.json" 2> "$O/tick.ring.err"; head -c 1800 $O/tick.ring.json
