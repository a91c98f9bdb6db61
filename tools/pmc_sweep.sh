#!/bin/bash
# FETCH_SIZE of k_fc_duo at full width (2500 pairs; 1 and 3 windows) with and without the workgroup's table timeline
# (DNE_DUO_SWEEP): one rocprofv3 --pmc pass each, kernel trace only.   gpurun -- 'bash tools/pmc_sweep.sh'
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for m in 0 1; do for ns in 1 3; do
DNE_DUO_SWEEP=$m DNE_NSUB=$ns rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/ps_${m}_${ns} -o kb -- python $R/tools/kbench.py --pairs 2500 --reps 1 --tslimit 6 > /dev/null 2>&1
f=$(find /tmp/ps_${m}_${ns} -name '*counter_collection.csv' | head -1)
python - "$f" "sweep=$m nsub=$ns" <<'PY'
import csv, collections, sys
by = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_fc_duo' in r['Kernel_Name']: by[int(r.get('Grid_Size', 0))][r['Dispatch_Id']] += float(r['Counter_Value'])
for k, v in by.items(): print(sys.argv[2], 'grid', k, 'dispatches', len(v), 'FETCH_SIZE_KB avg', sum(v.values())/len(v), '-> x2 GB', 2*sum(v.values())/len(v)/1e6)
PY
rm -rf /tmp/ps_${m}_${ns}
done; done
