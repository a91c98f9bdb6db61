#!/bin/bash
# k_fc_sub with the rolling window, the fused head and no wave priority: bit-exactness, GA width profile, ES mid-range A/B
TAG=${1:-r04e}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
R=$PWD
timeout 900 python -m pytest tests/test_gpu_edges.py -x -q -k "variant" > $O/pytest_variants.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_variants.log
for cfg in "DNE_FC_SUB=0" "X=0" "DNE_FC_SUB_HEAD=0" "DNE_FC_SUB_NSUB=3" "DNE_FC_SUB_PRIO=3" "DNE_FC_SUB_NSUB=1"; do
  env $cfg timeout 300 python tools/ga_lockstep_profile.py > $O/ga_prof.$cfg.json 2> $O/ga_prof.$cfg.err
  python - "$O/ga_prof.$cfg.json" "$cfg" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); g=d["generation_1"]
print(sys.argv[2], "gen1 %.1f ms %.0f steps/s" % (g["wall_ms"], g["steps_per_s"]), d["lock_step_us_at_width"])
PY
done
timeout 600 python tools/ab_inproc.py --pairs 312 --skip alone --gens 10 "X=0" "DNE_FC_SUB=2" "DNE_FC_SUB=2 DNE_FC_SUB_NSUB=3" "DNE_FC_SUB=2 DNE_FC_SUB_HEAD=0" > $O/ab_312.jsonl 2> $O/ab_312.err; tail -1 $O/ab_312.jsonl
timeout 600 python tools/ab_inproc.py --pairs 625 --skip alone --gens 10 "X=0" "DNE_FC_SUB=2" "DNE_FC_SUB=2 DNE_FC_SUB_NSUB=3" "DNE_FC_SUB=2 DNE_FC_SUB_NSUB=4" > $O/ab_625.jsonl 2> $O/ab_625.err; tail -1 $O/ab_625.jsonl
cd /tmp && export TMPDIR=/tmp
for w in 250 1000; do
  cfg="X=0"
  env $cfg rocprofv3 --kernel-trace --stats --output-format csv -d $O/st.$w -o g -- python $R/tools/ga_width_run.py $w 200 > $O/run.$w.json 2> $O/run.$w.err
  f=$(find $O/st.$w -name '*kernel_stats.csv' | head -1)
  echo "== width $w: $(cat $O/run.$w.json)"
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:7]:
    print("  %-60s calls %6s avg %8.1f us  %5s %%" % (r["Name"].replace("void ","").split("(")[0][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
  cp "$f" $O/kernel_stats.$w.csv; rm -rf $O/st.$w
done
