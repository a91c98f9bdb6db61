#!/bin/bash
# (experiment, not in the product tree: `git apply tools/experiments/ring_nw.patch && make -C deep-neuroevolution_amd/csrc` first)
# k_fc_ring with 8 / 10 / 11 / 12 compute waves per workgroup (DNE_RING_NW): smoke first (stop at a failure), then the same-box A/B
# (alone, lock-step, generations 3..8; the theta digests of all settings must be equal)
TAG=${1:-r05nw}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for nw in 11 12 10; do
  env DNE_RING_NW=$nw timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.$nw.log 2>&1 || { echo "smoke NW=$nw FAILED"; tail -8 $O/smoke.$nw.log | cut -c1-300; exit 1; }
  tail -1 $O/smoke.$nw.log | cut -c1-160
done
timeout 400 python tools/ab_inproc.py "X=0" "DNE_RING_NW=10" "DNE_RING_NW=11" "DNE_RING_NW=12" --rounds 2 --gens 6 > $O/ab.jsonl 2> $O/ab.err || { echo "ab FAILED"; tail -5 $O/ab.err; exit 1; }
tail -1 $O/ab.jsonl
