#!/bin/bash
# the rotated row loop of k_fc_ring (-DDNE_RING_ROT=1, ab/libdne_hip_rot.so): parity, then same-box A/B against the default library
TAG=${1:-r05q}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
ROT=$R/deep-neuroevolution_amd/csrc/ab/libdne_hip_rot.so
cd $R
DNE_LIB_PATH=$ROT timeout 300 python -m pytest tests/test_gpu_edges.py -x -q -k "test_every_step_kernel_variant and (knobs2- or knobs20 or knobs21 or knobs22 or knobs23)" > $O/pytest_ring.log 2>&1 || { echo "ring subset FAILED"; tail -15 $O/pytest_ring.log | cut -c1-300; exit 1; }
tail -1 $O/pytest_ring.log
DNE_LIB_PATH=$ROT timeout 400 python -m pytest tests/test_gpu_fullsize.py -x -q -k "test_full_generation_bit_exact" > $O/pytest_full.log 2>&1 || { echo "full FAILED"; tail -15 $O/pytest_full.log | cut -c1-300; exit 1; }
tail -1 $O/pytest_full.log
for lib in default rot; do
  P=""; [ $lib = rot ] && P=$ROT
  DNE_LIB_PATH=$P timeout 200 python tools/ab_inproc.py "X=0" --rounds 2 --gens 6 > $O/ab.$lib.jsonl 2> $O/ab.$lib.err || { echo "ab $lib FAILED"; tail -3 $O/ab.$lib.err; exit 1; }
  echo "$lib: $(tail -1 $O/ab.$lib.jsonl)"
done
