cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for duo in 0 1; do
  for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
    tag=$(echo $c | tr ' ' '_')
    DNE_FC_DUO=$duo DNE_NSUB=1 timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/duo${duo}_$tag -o kb -- python $GRAFT_REPO_ROOT/tools/kbench.py --reps 1 --tslimit 6 --sort-idx > $O/duo${duo}_$tag.log 2>&1
  done
done
python - <<'PY'
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT") + "/gpurun_out/r02v"
for d in sorted(glob.glob(O + "/duo*_*/")):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no csv"); continue
    by = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'k_fc' in k or 'k_out' in k or 'k_unit' in k:
            by[(k, r['Counter_Name'])][r['Dispatch_Id']] += float(r['Counter_Value'])
    for k, v in sorted(by.items()):
        print(os.path.basename(d.rstrip('/')), k, "n=%d avg=%.1f" % (len(v), sum(v.values()) / len(v)))
PY
