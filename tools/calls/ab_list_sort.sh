#!/bin/bash
# (experiment, not in the product tree: `git apply tools/experiments/list_sort.patch && make -C deep-neuroevolution_amd/csrc` first)
# the active list in noise-table order (DNE_LIST_SORT): smoke with the knob (stop at a failure), then the same-box A/B
# (alone, lock-step, generations 3..8; the theta digests of all settings must be equal)
TAG=${1:-r05ls}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
env DNE_LIST_SORT=1 timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo "smoke FAILED"; tail -8 $O/smoke.log | cut -c1-300; exit 1; }
tail -1 $O/smoke.log | cut -c1-160
timeout 400 python tools/ab_inproc.py ${AB_SETTINGS:-"X=0" "DNE_LIST_SORT=1" "DNE_NSUB_FULL=2" "DNE_LIST_SORT=1 DNE_FC_RING=2"} --rounds 2 --gens 6 > $O/ab.jsonl 2> $O/ab.err || { echo "ab FAILED"; tail -5 $O/ab.err; exit 1; }
tail -1 $O/ab.jsonl
