cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03b; mkdir -p $O
L=$O/kb11.log
for pairs in 2500 1800; do
for env in "DNE_DUO_SOLO_MAX=0" "DNE_DUO_SOLO_MAX=100000" "DNE_DUO_SOLO_MAX=100000 DNE_NSUB=2" "DNE_DUO_SOLO_MAX=100000 DNE_NSUB=4" "DNE_DUO_SOLO_MAX=0 DNE_NSUB=4"; do
  echo "== $pairs $env" >> $L
  env $env timeout 300 python tools/kbench.py --pairs $pairs --tslimit 24 --reps 3 2>&1 | grep rep | tail -2 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k: round(v,3) for k,v in d['per_step_ms'].items()}, round(d['step_wall_ms'],3))" >> $L
done
done
cat $L
