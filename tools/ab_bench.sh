#!/bin/bash
# Same-box A/B of the driver's command: every argument is one environment setting ("DNE_FC_DUO=0", "DNE_SPEC_MAX=0 DNE_NSUB=4", ...),
# "X=0" = the defaults.  Runs on one lease repeat to about +-0.1 %, different leases differ by +-2 %: compare within one call only.
#   gpurun -- 'bash tools/ab_bench.sh X=0 DNE_FC_DUO=0 X=0'
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=gpurun_out/${AB_TAG:-ab}; mkdir -p $O
for e in "$@"; do
  env $e timeout 900 python bench.py --gpus 1 --steps ${AB_STEPS:-20} --warmup ${AB_WARMUP:-5} --no-cpu-baseline --extra none ${AB_ARGS:-} > $O/b.json 2> $O/b.err
  echo "$e: $(tail -1 $O/b.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms/generation', round(d['value']), 'env-steps/s')")" | tee -a $O/ab.log
done
