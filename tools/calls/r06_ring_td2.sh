#!/bin/bash
# k_fc_ring with its base rows two ticks ahead (csrc/ab/libdne_hip_TD2.so, or whatever LIB names) against the product library:
# the ring's edge cases + the full-size generation on the candidate (stop at a failure), then alone / lock-step / generation same-box
set -u
TAG=${1:-r06m}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
LIB=${LIB:-$R/deep-neuroevolution_amd/csrc/ab/libdne_hip_TD2.so}
DNE_LIB_PATH=$LIB DNE_TEST_VARIANTS=1 timeout 300 python -m pytest tests/test_gpu_edges.py -x -q -k "test_every_step_kernel_variant and (knobs14 or knobs15 or knobs16 or knobs17 or knobs18)" > $O/pytest_ring.log 2>&1 || { echo "ring subset FAILED"; tail -15 $O/pytest_ring.log | cut -c1-300; exit 1; }
tail -1 $O/pytest_ring.log
DNE_LIB_PATH=$LIB timeout 400 python -m pytest tests/test_gpu_fullsize.py -x -q -k "test_full_generation_bit_exact or test_generations_past_zero" > $O/pytest_full.log 2>&1 || { echo "full FAILED"; tail -15 $O/pytest_full.log | cut -c1-300; exit 1; }
tail -1 $O/pytest_full.log
for lib in product cand product cand; do
  if [ $lib = cand ]; then export DNE_LIB_PATH=$LIB; else unset DNE_LIB_PATH; fi
  timeout 600 python tools/ab_inproc.py "X=0" --rounds 1 --gens ${GENS:-8} > $O/ab_$lib.jsonl 2> $O/ab_$lib.err
  tail -1 $O/ab_$lib.jsonl | python -c "
import json,sys
for k,v in json.loads(sys.stdin.read())['summary'].items(): print('$lib', k, 'alone', v.get('alone_fc_ms'), 'lockstep', v['lockstep_ms'], 'gen', v['gen_ms'], v['theta_sha'][:16])"
done
