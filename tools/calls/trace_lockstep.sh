#!/bin/bash
# kernel trace of six full-width lock-steps (4 windows), per setting of the knobs in SETTINGS (";"-separated): start / end of every kernel
TAG=${1:-r05tr}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
IFS=';' read -ra SET <<< "${SETTINGS:-X=0}"
for s in "${SET[@]}"; do
  env $s timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/t$i.d -o t -- python $R/tools/kbench.py --pairs 2500 --reps 1 --tslimit 6 > $O/t$i.json 2> $O/t$i.err
  f=$(find $O/t$i.d -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python - "$f" "$O/trace.$i.csv" "$s" <<'PY'
import csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('dne::', ''), r.get('Stream_Id', r.get('Queue_Id', ''))) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(); t0 = rows[0][0]
with open(sys.argv[2], 'w') as f:
    f.write("# %s\nstart_us,end_us,kernel,queue\n" % sys.argv[3])
    for a, b, k, q in rows: f.write("%.1f,%.1f,%s,%s\n" % ((a - t0) / 1e3, (b - t0) / 1e3, k[:40], q))
PY
  rm -rf $O/t$i.d; i=$((i+1))
done
ls -la $O
