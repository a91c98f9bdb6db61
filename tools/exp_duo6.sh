cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02z; mkdir -p $O
L=$O/kbench6.log
for env in "DNE_FC_GRID=256" "DNE_FC_GRID=256 DNE_NSUB=2" "DNE_FC_GRID=256 DNE_NSUB=4" "DNE_FC_GRID=320" "DNE_FC_GRID=384" "DNE_FC_GRID=192" "DNE_FC_GRID=256 DNE_NSUB=6"; do
  echo "== $env" >> $L
  env $env timeout 300 python tools/kbench.py --tslimit 24 --reps 2 --sort-idx 2>&1 | grep rep | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['per_step_ms'], d['step_wall_ms'])" >> $L
done
cat $L
