#!/bin/bash
# the driver's round-end GPU tier on the final tree: the GPU suite, then smoke()
TAG=${1:-r05t}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=10 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -22 $O/pytest_gpu.log | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-300
