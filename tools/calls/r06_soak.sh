#!/bin/bash
# Flakiness check of what the driver runs at round end: the GPU suite twice, smoke twice, the driver's bench command twice (headline only)
TAG=${1:-r06soak}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for i in 1 2; do
  timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_$i.log 2>&1; echo "pytest $i rc=$? $(tail -1 $O/pytest_$i.log)"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_$i.log 2>&1; echo "smoke $i rc=$? $(tail -1 $O/smoke_$i.log | cut -c1-120)"
  ( cd /tmp && timeout 600 python $R/bench.py --steps 20 --warmup 5 --extra none > $O/bench_$i.json 2> $O/bench_$i.err ); python -c "
import json; d=json.loads([l for l in open('$O/bench_$i.json') if l.startswith('{')][-1]); print('bench $i', round(d['value']), round(d['ms_per_step'],2), d['roofline']['frac'])"
done
