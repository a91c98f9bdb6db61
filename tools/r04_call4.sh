#!/bin/bash
# per-kernel durations of a Deep-GA lock-step at fixed width, old mid-range kernels vs k_fc_sub
TAG=${1:-r04d}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for w in 250 1000; do
for cfg in "DNE_FC_SUB=0" "DNE_FC_SUB=1"; do
  env $cfg rocprofv3 --kernel-trace --stats --output-format csv -d $O/st.$w.$cfg -o g -- python $R/tools/ga_width_run.py $w 200 > $O/run.$w.$cfg.json 2> $O/run.$w.$cfg.err
  f=$(find $O/st.$w.$cfg -name '*kernel_stats.csv' | head -1)
  echo "== width $w $cfg: $(cat $O/run.$w.$cfg.json)"
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("  %-60s calls %6s avg %8.1f us  %5s %%" % (r["Name"].replace("void ","").split("(")[0][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
  cp "$f" $O/kernel_stats.$w.$cfg.csv; rm -rf $O/st.$w.$cfg
done; done
