#!/bin/bash
# The renderer's vertical pass on quads of output pixels (env_synth.h) against the tree before it (csrc/ab/libdne_hip_base.so, built from the
# parent commit before the call): parity first, then the in-kernel phase clock, the full-width lock-step and generations 3-8 of the driver's workload
#   gpurun -- 'bash tools/calls/r06_render_quads.sh r06q1'
set -u
TAG=${1:-r06q1}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
C=$R/deep-neuroevolution_amd/csrc
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -x -q -m gpu 2>&1 | grep -v '^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version' | tail -3 | tee $O/parity.txt
grep -q failed $O/parity.txt && exit 1
DNE_LIB_PATH=$C/libdne_hip_clock.so timeout 200 python tools/render_phase_clock.py 2500 2>&1 | tail -1 | tee $O/render_phase.json
for round in 1 2; do
  for v in base new; do
    lib=$C/ab/libdne_hip_$v.so; [ $v = new ] && lib=$C/libdne_hip.so
    echo "== $v"; DNE_LIB_PATH=$lib timeout 200 python tools/kbench.py --pairs 2500 --reps 2 --tslimit 16 2>&1 | grep '"rep": 1' | tee -a $O/kbench.$v.json | cut -c1-330
    DNE_LIB_PATH=$lib timeout 400 python tools/ab_inproc.py "X=0" --rounds 1 --gens 6 --skip alone,lockstep 2>$O/ab.$v.err | tail -1 | python -c "
import json,sys
for k,v in json.loads(sys.stdin.read())['summary'].items(): print('$v gen_ms', v['gen_ms'], v['theta_sha'][0][:8])" | tee -a $O/gen.$v.txt
  done
done
