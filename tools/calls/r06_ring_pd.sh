#!/bin/bash
# k_fc_ring: prefetch distance 6 instead of 5 (one more tick for a noise segment to land), cache policies of the ring's DMAs, LDS-DMA beyond 64 KB
set -u
TAG=${1:-r06l}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/micro/lds_dma_hi | tee $O/lds_dma_hi.json
for v in ${CHECK:-PD6}; do
  DNE_LIB_PATH=$R/deep-neuroevolution_amd/csrc/ab/libdne_hip_$v.so timeout 600 python -m pytest tests/test_gpu_edges.py -m gpu -k "ring" -x -q 2>&1 | tail -3
done
VARIANTS="${VARIANTS:-default PD6 PD6_NT SC1 SC01 SC01NT PD6_SC1}" bash tools/calls/r06_ring_hot.sh $TAG
