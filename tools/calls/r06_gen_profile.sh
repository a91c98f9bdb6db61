#!/bin/bash
# Round 6: where a TRAINED generation's milliseconds go, by active width, for one rank of 1 / 2 / 4 / 8 (tools/gen_profile.py) + parity of the renderer with
# the trimmed resize tables + its alone-time
TAG=${1:-r06e}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python tools/gen_profile.py --gens 15 --worlds 1,2,4,8 > $O/gen_profile.jsonl 2> $O/gen_profile.err; cat $O/gen_profile.jsonl | cut -c1-1200
DNE_NSUB=1 timeout 300 python tools/kbench.py --pairs 2500 --reps 2 --tslimit 12 > $O/kbench_alone.json 2>&1; tail -4 $O/kbench_alone.json | cut -c1-400
timeout 600 python tools/ab_inproc.py "X=0" "Y=0" --rounds 2 --gens 6 --skip alone > $O/ab.jsonl 2> $O/ab.err; tail -1 $O/ab.jsonl | cut -c1-600
