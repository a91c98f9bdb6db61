cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_edges.py -x -q -m gpu -k "variant" > $O/pytest_variants.log 2>&1; tail -3 $O/pytest_variants.log
L=$O/kbench.log
for env in "DNE_NSUB=1" "DNE_NSUB=1 DNE_FC_GRID=256" "DNE_NSUB=1 DNE_FC_GRID=417" "DNE_NSUB=1 DNE_FC_DUO=0" "DNE_FC_DUO=1" "DNE_FC_DUO=0" "DNE_FC_DUO=1 DNE_NSUB=2" "DNE_FC_DUO=1 DNE_NSUB=4"; do
  echo "== $env" >> $L
  env $env timeout 300 python tools/kbench.py --tslimit 24 --reps 2 --sort-idx 2>&1 | grep rep | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['per_step_ms'], d['step_wall_ms'])" >> $L
done
cat $L
