#!/bin/bash
# What k_fc_ring's tick waits for: timing-only builds (WRONG numerics) whose base rows / noise segments always hit the caches
#   gpurun -- 'bash tools/calls/r06_ring_hot.sh r06j'   (libs built before the call: csrc/ab/libdne_hip_{THETA_HOT,DMA_HOT,BOTH_HOT}.so)
set -u
TAG=${1:-r06j}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for v in ${VARIANTS:-default THETA_HOT DMA_HOT BOTH_HOT}; do
  lib=$R/deep-neuroevolution_amd/csrc/ab/libdne_hip_$v.so; [ $v = default ] && lib=$R/deep-neuroevolution_amd/csrc/libdne_hip.so
  for ns in 1 4; do
    echo "== $v nsub=$ns"; DNE_LIB_PATH=$lib DNE_NSUB=$ns DNE_NSUB_FULL=$ns timeout 200 python tools/kbench.py --pairs 2500 --reps 2 --tslimit 16 2>&1 | grep '"rep": 1' | tee -a $O/$v.nsub$ns.json | cut -c1-330
  done
done
