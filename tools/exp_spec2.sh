cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for env in "DNE_SPEC_MAX=8" "DNE_SPEC_MAX=0"; do
rm -rf $O/tr; env $env timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o t -- python $GRAFT_REPO_ROOT/tools/tail_bench.py 1 > /dev/null 2>&1
echo "== $env"
python - <<PY
import csv, glob, collections
f = glob.glob("$O/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# take the last 2000 dispatches (tail steps)
rows = rows[-1500:]
d = collections.defaultdict(list)
for r in rows:
    d[r['Kernel_Name'].split('(')[0].replace('void ','')[:60]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    print("%-62s n=%4d avg=%6.2f us" % (k, len(v), sum(v)/len(v)))
gaps=[(int(rows[i+1]['Start_Timestamp'])-int(rows[i]['End_Timestamp']))/1e3 for i in range(len(rows)-1)]
print("avg gap us", sum(gaps)/len(gaps))
PY
done
