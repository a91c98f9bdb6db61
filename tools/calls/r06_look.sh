#!/bin/bash
# Round 6, first look: what the window chains look like away from full width and why k_conv12 is slow beside k_fc_ring.
#   1. smoke  2. kernel traces of fixed-width lock-steps at 2500 / 1250 / 625 / 312 ES pairs and 1000 / 250 GA children (durations AND gaps per stream)
#   3. lock-steps by width (len_profile) at the four shares  4. the per-workgroup clock (profiling build): k_conv12 / k_out / render alone vs in the mix
#   5. hardware queues: GPU_MAX_HW_QUEUES A/B on the bench  6. k_fc_ring in the sparse range at a rank's share (DNE_FC_RING=2)
TAG=${1:-r06a}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo "smoke FAILED"; tail -8 $O/smoke.log | cut -c1-300; exit 1; }
tail -1 $O/smoke.log | cut -c1-160
cd /tmp; export TMPDIR=/tmp
trace() {  # label command...
  local lab=$1; shift
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/$lab.d -o t -- "$@" > $O/$lab.run.json 2> $O/$lab.err
  f=$(find $O/$lab.d -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python $R/tools/trace_summary.py "$f" "$lab" --csv $O/$lab.trace.csv > $O/$lab.summary.json 2>> $O/$lab.err
  rm -rf $O/$lab.d
  head -c 600 $O/$lab.summary.json; echo
}
for p in 2500 1250 625 312; do trace es_$p python $R/tools/kbench.py --pairs $p --reps 1 --tslimit 8; done
for m in 1000 250; do trace ga_$m python $R/tools/ga_kbench.py --members $m --reps 1 --tslimit 8; done
for p in 2500 1250 625 312; do timeout 200 python $R/tools/len_profile.py --pairs $p > $O/len_profile_$p.json 2> $O/len_profile_$p.err; done
CLK=$R/deep-neuroevolution_amd/csrc/libdne_hip_clock.so
DNE_LIB_PATH=$CLK timeout 300 python $R/tools/wg_clock.py "DNE_NSUB=1" "X=0" "DNE_FC_PRIO=0" > $O/wg_clock_2500.jsonl 2> $O/wg_clock_2500.err
DNE_LIB_PATH=$CLK timeout 300 python $R/tools/wg_clock.py "DNE_NSUB=1" "X=0" --pairs 625 > $O/wg_clock_625.jsonl 2> $O/wg_clock_625.err
cut -c1-1500 $O/wg_clock_2500.jsonl
B="python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --extra none"
for q in default 2 8; do
  if [ $q = default ]; then $B 2>/dev/null | tail -1 > $O/queues_$q.json; else GPU_MAX_HW_QUEUES=$q $B 2>/dev/null | tail -1 > $O/queues_$q.json; fi
  python -c "import json,sys; d=json.load(open('$O/queues_$q.json')); print('queues $q', d['value'], d['ms_per_step'])"
done
GPU_MAX_HW_QUEUES=8 DNE_NSUB_FULL=6 $B 2>/dev/null | tail -1 > $O/queues_8_nsub6.json
python -c "import json,sys; d=json.load(open('$O/queues_8_nsub6.json')); print('queues 8 nsub 6', d['value'], d['ms_per_step'])"
cd $R
for p in 1250 625; do
  timeout 400 python tools/ab_inproc.py "X=0" "DNE_FC_RING=2" --pairs $p --rounds 2 --gens 6 --skip alone > $O/ab_ring2_$p.jsonl 2> $O/ab_ring2_$p.err
  tail -1 $O/ab_ring2_$p.jsonl | cut -c1-700
done
ls $O
