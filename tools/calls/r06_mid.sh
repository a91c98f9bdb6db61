#!/bin/bash
# Round 6, VERDICT round 5 items 1a / 1b / 2 / 4 on the product library:
#   1. parity of the kernel-level tests (the table-ordered list is a schedule: results must not move)
#   2. kernel traces of fixed-width lock-steps (durations AND gaps per stream): 2500 / 1250 / 625 / 312 ES pairs, 1000 / 250 GA children,
#      plus 2500 pairs in ONE window = every kernel alone (tools/alone_times.py reads that one)
#   3. the ring on a table-ordered list below 1500 pairs (DNE_LIST_SORT + DNE_FC_RING=2), same-process A/B at a rank's share of 1250 / 625 pairs
#   4. counters of k_conv12 / k_out / k_env_render (rocprofv3 --pmc serialises dispatches: these are the kernels ALONE; beside the ring: tools/wg_clock.py)
TAG=${1:-r06c}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
cd /tmp; export TMPDIR=/tmp
trace() {  # label command...
  local lab=$1; shift
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/$lab.d -o t -- "$@" > $O/$lab.run.json 2> $O/$lab.err
  f=$(find $O/$lab.d -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python $R/tools/trace_summary.py "$f" "$lab" --csv $O/$lab.trace.csv > $O/$lab.summary.json 2>> $O/$lab.err
  rm -rf $O/$lab.d
}
for p in 2500 1250 625 312; do trace es_$p python $R/tools/kbench.py --pairs $p --reps 1 --tslimit 8; done
DNE_NSUB=1 trace es_2500_alone python $R/tools/kbench.py --pairs 2500 --reps 1 --tslimit 8
for m in 1000 250; do trace ga_$m python $R/tools/ga_kbench.py --members $m --reps 1 --tslimit 8; done
DNE_LIST_SORT=1 DNE_FC_RING=2 trace es_1250_ring_sorted python $R/tools/kbench.py --pairs 1250 --reps 1 --tslimit 8
DNE_LIST_SORT=1 DNE_FC_RING=2 trace es_625_ring_sorted python $R/tools/kbench.py --pairs 625 --reps 1 --tslimit 8
python - <<PY
import json
for lab in ("es_2500", "es_2500_alone", "es_1250", "es_1250_ring_sorted", "es_625", "es_625_ring_sorted", "es_312", "ga_1000", "ga_250"):
    try: d = json.load(open("$O/%s.summary.json" % lab))
    except Exception as e: print(lab, "no summary", e); continue
    print("==", lab, "span", d["lock_step_span_us"], "sum/span", d["sum_of_durations_over_span"], {s: v["period_us_median"] for s, v in d["streams"].items()})
    for k, v in d["kernels"].items(): print("   %-40s n=%4d dur %7.1f gap %s" % (k[:40], v["launches"], v["dur_us_mean"], v["gap_before_us_mean"]))
PY
cd $R
for p in 1250 625; do
  timeout 600 python tools/ab_inproc.py "X=0" "DNE_LIST_SORT=1" "DNE_LIST_SORT=1 DNE_FC_RING=2" "DNE_LIST_SORT=1 DNE_FC_RING=2 DNE_NSUB=2" "DNE_LIST_SORT=1 DNE_FC_RING=2 DNE_NSUB=3" \
      --pairs $p --rounds 2 --gens 6 --skip alone > $O/ab_ring_sorted_$p.jsonl 2> $O/ab_ring_sorted_$p.err
  tail -1 $O/ab_ring_sorted_$p.jsonl | python -c "
import json,sys
for k,v in json.loads(sys.stdin.read())['summary'].items(): print('$p', k, v['lockstep_ms'], v['gen_ms'], v['theta_sha'])"
done
cd /tmp
reduce() {  # counter_collection.csv -> per (kernel, counter): dispatches, sum
  python - "$1" "$2" <<'PY'
import csv, collections, sys
tot = collections.defaultdict(float); disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = (r['Kernel_Name'].split('(')[0].replace('void ', ''), r['Counter_Name'])
    tot[k] += float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
with open(sys.argv[2], 'w') as f:
    f.write("kernel,counter,dispatches,sum\n")
    for k in sorted(tot):
        f.write('"%s",%s,%d,%.1f\n' % (k[0], k[1], len(disp[k]), tot[k]))
PY
}
pmc() {  # name "counters" command...
  local name=$1 ctr=$2; shift 2
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$O/$name.d" -o p -- "$@" > "$O/$name.json" 2> "$O/$name.err"
  f=$(find "$O/$name.d" -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then reduce "$f" "$O/$name.csv"; else echo "no counter file for $name" >> "$O/errors.log"; tail -5 "$O/$name.err" >> "$O/errors.log"; fi
  rm -rf "$O/$name.d"
}
K="python $R/tools/kbench.py --pairs 2500 --reps 1 --tslimit 6"
pmc pmc_A "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" $K
pmc pmc_B "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" $K
pmc pmc_C "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" $K
pmc pmc_D "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" $K
pmc pmc_E "SPI_RA_LDS_CU_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_REQ_NO_ALLOC_CSN" $K
pmc pmc_F "FETCH_SIZE" $K
pmc pmc_G "WRITE_SIZE" $K
cat $O/errors.log 2>/dev/null | head -20
ls $O | wc -l
