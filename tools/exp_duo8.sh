cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02z; mkdir -p $O
L=$O/mid.log
for p in 312 625 1250; do
 for env in "DNE_FC_DUO_MIN=800" "DNE_FC_DUO_MIN=97" "DNE_FC_DUO_MIN=97 DNE_FC_GRID=256"; do
  echo "== pairs $p $env" >> $L
  env $env timeout 300 python tools/mid_bench.py $p 40 2>&1 | tail -1 >> $L
 done
done
cat $L
timeout 900 python bench.py --steps 6 --warmup 2 > $O/bench_duo.json 2> $O/bench_duo.err; tail -1 $O/bench_duo.json | cut -c1-600
DNE_FC_DUO=0 timeout 900 python bench.py --steps 6 --warmup 2 > $O/bench_fc2.json 2> $O/bench_fc2.err; tail -1 $O/bench_fc2.json | cut -c1-300
