cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02x; mkdir -p $O
L=$O/kbench.log
for env in "DNE_FC_DUO=1" "DNE_DEBUG_SKIP=8" "DNE_DEBUG_SKIP=16" "DNE_DEBUG_SKIP=24" "DNE_FC_GRID=256" "DNE_FC_GRID=768" "DNE_FC_GRID=1024" "DNE_FC_RB=2 DNE_FC_GRID=768" "DNE_FC_RB=2 DNE_FC_GRID=1024" "DNE_FC_DUO=0 DNE_FC_GRID=256" "DNE_FC_DUO=0 DNE_FC_GRID=768"; do
  echo "== NSUB=1 $env" >> $L
  env $env DNE_NSUB=1 timeout 300 python tools/kbench.py --tslimit 24 --reps 2 --sort-idx 2>&1 | grep rep | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['per_step_ms'], d['step_wall_ms'])" >> $L
done
cat $L
