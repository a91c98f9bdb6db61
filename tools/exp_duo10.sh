cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03b; mkdir -p $O
L=$O/mid.log
for p in 200 312 625 1250; do
 for env in "DNE_FC_DUO_MIN=800" "DNE_FC_DUO_MIN=97 DNE_DUO_SOLO_MAX=100000" "DNE_FC_DUO_MIN=97 DNE_DUO_SOLO_MAX=100000 DNE_NSUB=2" "DNE_FC_DUO_MIN=97 DNE_DUO_SOLO_MAX=100000 DNE_NSUB=1"; do
  echo "== pairs $p $env" >> $L
  env $env timeout 300 python tools/mid_bench.py $p 40 2>&1 | tail -1 >> $L
 done
done
cat $L
