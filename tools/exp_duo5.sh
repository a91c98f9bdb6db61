cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02z; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for env in "DNE_FC_GRID=512" "DNE_FC_GRID=256" "DNE_FC_GRID=128" "DNE_FC_GRID=256 DNE_DUO_LAG=1"; do
    i=$((i+1))
    env $env DNE_NSUB=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/v$i -o kb -- python $GRAFT_REPO_ROOT/tools/kbench.py --reps 1 --tslimit 6 --sort-idx > $O/v$i.log 2>&1
    echo "v$i = $env" >> $O/key.txt
done
cat $O/key.txt
python - <<'PY'
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT") + "/gpurun_out/r02z"
for d in sorted(glob.glob(O + "/v*/")):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    by = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'k_fc_duo' in k:
            by[(k, r['Counter_Name'])][r['Dispatch_Id']] += float(r['Counter_Value'])
    ts = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    dur = [ (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(ts[0])) if 'k_fc_duo' in r['Kernel_Name']]
    for k, v in sorted(by.items()):
        print(os.path.basename(d.rstrip('/')), k, "n=%d avg=%.1f" % (len(v), sum(v.values()) / len(v)), "dur_us(avg, under pmc)=%.0f" % (sum(dur)/max(len(dur),1)))
PY
