#!/bin/bash
# profiles/r06_alone_times.json on the final tree: two kernel traces of 8 full-width lock-steps (one window = every kernel alone; the product schedule)
TAG=${1:-r06w}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
trace() {  # label env... -- command...
  local lab=$1; shift
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/$lab.d -o t -- "$@" > $O/$lab.run.json 2> $O/$lab.err
  f=$(find $O/$lab.d -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python $R/tools/trace_summary.py "$f" "$lab" > $O/$lab.summary.json 2>> $O/$lab.err
  rm -rf $O/$lab.d
}
DNE_NSUB=1 trace es_2500_alone python $R/tools/kbench.py --pairs 2500 --reps 1 --tslimit 8
trace es_2500 python $R/tools/kbench.py --pairs 2500 --reps 1 --tslimit 8
python $R/tools/alone_times.py $O/es_2500_alone.summary.json $O/es_2500.summary.json | tee $O/alone_times.json | cut -c1-600
