cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02z; mkdir -p $O
L=$O/kbench7.log
for env in "DNE_FC_GRID=256 DNE_NSUB=2" "DNE_FC_GRID=224 DNE_NSUB=2" "DNE_FC_GRID=288 DNE_NSUB=2" "DNE_FC_GRID=256 DNE_NSUB=3" "DNE_FC_GRID=512 DNE_NSUB=2" "DNE_FC_GRID=256 DNE_NSUB=2 DNE_CONV_FUSED=0" "DNE_FC_DUO=0 DNE_NSUB=2 DNE_FC_GRID=256" "DNE_FC_DUO=0 DNE_NSUB=3"; do
  echo "== $env" >> $L
  env $env timeout 300 python tools/kbench.py --tslimit 24 --reps 4 --sort-idx 2>&1 | grep rep | tail -3 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k: round(v,3) for k,v in d['per_step_ms'].items()}, round(d['step_wall_ms'],3))" >> $L
done
cat $L
