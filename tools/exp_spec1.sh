cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py::test_full_generation_bit_exact > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
for env in "DNE_SPEC_MAX=8"; do
  echo "== $env"; env $env timeout 300 python tools/tail_bench.py 1,2,4,8 2>&1 | tail -1
done | tee $O/tail.log
cd /tmp && export TMPDIR=/tmp
rm -rf $O/tr; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o t -- python $GRAFT_REPO_ROOT/tools/tail_bench.py 1 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$O/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-900:]
d = collections.defaultdict(list)
for r in rows:
    d[r['Kernel_Name'].split('(')[0].replace('void ','')[:60]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:6]:
    print("%-62s n=%4d avg=%6.2f us" % (k, len(v), sum(v)/len(v)))
PY
