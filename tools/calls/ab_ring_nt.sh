#!/bin/bash
# nt vs plain ring DMAs: same-box, one process per library
TAG=${1:-r05i}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 120 python -m pytest tests/test_gpu_edges.py -x -q -k "test_every_step_kernel_variant and (knobs2- or knobs20 or knobs21 or knobs22 or knobs23)" > $O/pytest_ring.log 2>&1 || { echo "ring subset FAILED"; tail -15 $O/pytest_ring.log | cut -c1-300; exit 1; }
tail -1 $O/pytest_ring.log
for lib in nt plain; do
  P=""; [ $lib = plain ] && P=$R/deep-neuroevolution_amd/csrc/ab/libdne_hip_plain.so
  DNE_LIB_PATH=$P timeout 200 python tools/ab_inproc.py "X=0" --rounds 2 --gens 6 > $O/ab.$lib.jsonl 2> $O/ab.$lib.err || { echo "ab $lib FAILED"; tail -3 $O/ab.$lib.err; exit 1; }
  echo "$lib: $(tail -1 $O/ab.$lib.jsonl)"
done
cd /tmp; export TMPDIR=/tmp
for lib in nt plain; do
  P=""; [ $lib = plain ] && P=$R/deep-neuroevolution_amd/csrc/ab/libdne_hip_plain.so
  for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
    n=$(echo $grp | cut -d' ' -f1)
    DNE_LIB_PATH=$P DNE_NSUB=1 timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p.$lib.$n -o p -- python $R/tools/kbench.py --pairs 2500 --reps 1 --tslimit 6 > /dev/null 2> $O/p.$lib.$n.err
    f=$(find $O/p.$lib.$n -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python - "$f" "$lib" <<'PY'
import csv, collections, sys
tot = collections.defaultdict(float); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_fc_ring' in r['Kernel_Name']:
        tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']].add(r['Dispatch_Id'])
print(sys.argv[2], {k: round(v / max(len(n[k]), 1)) for k, v in tot.items()})
PY
    rm -rf $O/p.$lib.$n
  done
done
