#!/bin/bash
TAG=${1:-r05f}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
CLK=$R/deep-neuroevolution_amd/csrc/libdne_hip_clock.so
cd $R
timeout 600 python -m pytest tests/test_gpu_edges.py -x -q -k "test_reference_pass_under or test_every_step_kernel_variant" > $O/pytest_edges.log 2>&1; echo "edges rc=$?"; tail -3 $O/pytest_edges.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "test_full_generation_bit_exact" > $O/pytest_full_ring.log 2>&1; echo "full ring rc=$?"; tail -3 $O/pytest_full_ring.log
timeout 600 python tools/ab_inproc.py $AB_SETTINGS --rounds 2 > $O/ab.jsonl 2> $O/ab.err; echo "ab rc=$?"; tail -1 $O/ab.jsonl
env DNE_LIB_PATH=$CLK DNE_NSUB=1 timeout 200 python tools/duo_tick_clock.py > "$O/tick.ring.json" 2> "$O/tick.ring.err"; head -c 1200 $O/tick.ring.json
if [ "${PMC:-0}" = "1" ]; then MIX=0 FULL=0 bash tools/collect_pmc_duo_mem.sh $TAG > $O/pmc.log 2>&1; python tools/summarize_pmc_duo_mem.py $O/pmc_duo_mem r05 > $O/pmc_summary.txt 2>&1; cp profiles/r05_pmc_fc_ring_mem.json $O/ 2>/dev/null; tail -50 $O/pmc_summary.txt; fi
