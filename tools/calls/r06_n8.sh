#!/bin/bash
# Round 6: a rank of eight on its own stretch of the table (313 dense pairs): the sub-slice regime's knobs, and where its generation goes
TAG=${1:-r06i}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 600 python tools/gen_profile.py --gens 15 --worlds 2,4,8 > $O/gen_profile_table.jsonl 2> $O/gen_profile.err; cut -c1-900 $O/gen_profile_table.jsonl
timeout 900 python tools/shard_ab.py --worlds 8 --gens 6 --warmup 3 "table" "table DNE_FC_SUB_NSUB=1" "table DNE_FC_SUB_NSUB=2" "table DNE_FC_SUB_NSUB=4" \
   "table DNE_FC_SUB_SPW=2" "table DNE_FC_SUB_MAX=200" "table DNE_FC_SUB_HEAD=0" "table DNE_RENDER_THREADS=256" "table DNE_RENDER_THREADS=1024" "table DNE_BURST=16" "table DNE_CONV_FUSED=0" > $O/shard_ab.jsonl 2> $O/shard_ab.err
python - <<PY
import json
for l in open("$O/shard_ab.jsonl"):
    d = json.loads(l); print(d["world"], "%-40s" % d["setting"], d["ms_per_generation"], d["rank0_ms"], d["theta_sha"])
PY
cd /tmp; export TMPDIR=/tmp
trace() {  # label command...
  local lab=$1; shift
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/$lab.d -o t -- "$@" > $O/$lab.run.json 2> $O/$lab.err
  f=$(find $O/$lab.d -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python $R/tools/trace_summary.py "$f" "$lab" --csv $O/$lab.trace.csv > $O/$lab.summary.json 2>> $O/$lab.err
  rm -rf $O/$lab.d
}
trace es_313_dense python $R/tools/kbench.py --pairs 313 --reps 1 --tslimit 8 --idx-range 30000000
trace es_625_dense python $R/tools/kbench.py --pairs 625 --reps 1 --tslimit 8 --idx-range 62000000
python - <<PY
import json
for lab in ("es_313_dense", "es_625_dense"):
    try: d = json.load(open("$O/%s.summary.json" % lab))
    except Exception as e: print(lab, "no summary", e); continue
    print("==", lab, "span", d["lock_step_span_us"], "sum/span", d["sum_of_durations_over_span"], {s: v["period_us_median"] for s, v in d["streams"].items()})
    for k, v in d["kernels"].items(): print("   %-40s n=%4d dur %7.1f gap %s" % (k[:40], v["launches"], v["dur_us_mean"], v["gap_before_us_mean"]))
PY
