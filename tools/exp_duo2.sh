cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02w; mkdir -p $O
L=$O/kbench.log
for env in "DNE_DUO_LAG=0" "DNE_DUO_LAG=1" "DNE_DUO_LAG=2" "DNE_DUO_LAG=1 DNE_FC_RB=2" "DNE_DUO_LAG=2 DNE_FC_RB=2" "DNE_DUO_LAG=4 DNE_FC_RB=2"; do
  echo "== NSUB=1 $env" >> $L
  env $env DNE_NSUB=1 timeout 300 python tools/kbench.py --tslimit 24 --reps 2 --sort-idx 2>&1 | grep rep | tail -1 >> $L
  echo "== $env" >> $L
  env $env timeout 300 python tools/kbench.py --tslimit 24 --reps 2 --sort-idx 2>&1 | grep rep | tail -1 >> $L
done
cat $L
cd /tmp && export TMPDIR=/tmp
for lag in 1 2; do
    DNE_DUO_LAG=$lag DNE_NSUB=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/lag${lag} -o kb -- python $GRAFT_REPO_ROOT/tools/kbench.py --reps 1 --tslimit 6 --sort-idx > $O/lag${lag}.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT") + "/gpurun_out/r02w"
for d in sorted(glob.glob(O + "/lag*/")):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    by = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'k_fc_duo' in k:
            by[(k, r['Counter_Name'])][r['Dispatch_Id']] += float(r['Counter_Value'])
    for k, v in sorted(by.items()):
        print(os.path.basename(d.rstrip('/')), k, "n=%d avg=%.1f" % (len(v), sum(v.values()) / len(v)))
PY
