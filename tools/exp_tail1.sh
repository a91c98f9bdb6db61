cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03b; mkdir -p $O
for env in "X=1" "DNE_CONV_FUSED_MIN=1" "DNE_CONV_SPLIT_MAX=0" "DNE_CONV_FUSED_MIN=1 DNE_RENDER_BANDS=7"; do
  echo "== $env"; env $env timeout 300 python tools/tail_bench.py 1,4,16 2>&1 | tail -1
done | tee $O/tail1.log
