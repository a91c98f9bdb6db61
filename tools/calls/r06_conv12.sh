#!/bin/bash
# Round 6, item 2: k_conv12 with a 126-register footprint (was 200) -- does it now share CUs with k_fc_ring, and what does the generation gain?
#   1. parity (the kernel-level tests + the ring's edge cases)  2. workgroup clock  3. same-box A/B against round 5's library (csrc/ab/libdne_hip_r05.so)
#   4. population shares, both libraries  5. kernel traces of fixed-width lock-steps
TAG=${1:-r06b}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
cd /tmp; export TMPDIR=/tmp
CLK=$R/deep-neuroevolution_amd/csrc/libdne_hip_clock.so
OLD=$R/deep-neuroevolution_amd/csrc/ab/libdne_hip_r05.so
DNE_LIB_PATH=$CLK timeout 300 python $R/tools/wg_clock.py "DNE_NSUB=1" "X=0" > $O/wg_clock_2500.jsonl 2> $O/wg_clock_2500.err
DNE_LIB_PATH=$CLK timeout 300 python $R/tools/wg_clock.py "X=0" --pairs 625 > $O/wg_clock_625.jsonl 2> $O/wg_clock_625.err
python - <<PY
import json
for f in ("$O/wg_clock_2500.jsonl", "$O/wg_clock_625.jsonl"):
    for l in open(f):
        d = json.loads(l); print(d["setting"], d["pairs"])
        for k, v in d["kernels"].items(): print("   ", k, {a: b for a, b in v.items() if a in ("wg_us_mean", "busy_us_total", "resident_wgs_per_cu_while_running", "beside_streaming_fc", "cu_without_streaming_fc")})
PY
B="python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --extra none"
for rnd in 1 2; do
  for lib in old new; do
    if [ $lib = old ]; then DNE_LIB_PATH=$OLD $B 2>/dev/null | tail -1 > $O/bench_${lib}_$rnd.json; else $B 2>/dev/null | tail -1 > $O/bench_${lib}_$rnd.json; fi
    python -c "import json; d=json.load(open('$O/bench_${lib}_$rnd.json')); print('$lib $rnd', round(d['value']), round(d['ms_per_step'],2), d.get('theta_abs_sum_after'))"
  done
done
for p in 2500 1250 624; do
  for lib in old new; do
    if [ $lib = old ]; then DNE_LIB_PATH=$OLD $B --pop $p 2>/dev/null | tail -1 > $O/share_${lib}_$p.json; else $B --pop $p 2>/dev/null | tail -1 > $O/share_${lib}_$p.json; fi
    python -c "import json; d=json.load(open('$O/share_${lib}_$p.json')); print('pop $p $lib', round(d['value']), round(d['ms_per_step'],2))"
  done
done
trace() {  # label command...
  local lab=$1; shift
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/$lab.d -o t -- "$@" > $O/$lab.run.json 2> $O/$lab.err
  f=$(find $O/$lab.d -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python $R/tools/trace_summary.py "$f" "$lab" --csv $O/$lab.trace.csv > $O/$lab.summary.json 2>> $O/$lab.err
  rm -rf $O/$lab.d
}
for p in 2500 1250 625 312; do trace es_$p python $R/tools/kbench.py --pairs $p --reps 1 --tslimit 8; done
DNE_LIB_PATH=$OLD trace es_2500_r05 python $R/tools/kbench.py --pairs 2500 --reps 1 --tslimit 8
python - <<PY
import json
for lab in ("es_2500_r05", "es_2500", "es_1250", "es_625", "es_312"):
    try: d = json.load(open("$O/%s.summary.json" % lab))
    except Exception as e: print(lab, "no summary", e); continue
    print("==", lab, "span", d["lock_step_span_us"], "sum/span", d["sum_of_durations_over_span"], {s: (v["queue"], v["period_us_median"]) for s, v in d["streams"].items()})
    for k, v in d["kernels"].items(): print("   %-40s n=%4d dur %7.1f gap %s" % (k, v["launches"], v["dur_us_mean"], v["gap_before_us_mean"]))
PY
ls $O | wc -l
