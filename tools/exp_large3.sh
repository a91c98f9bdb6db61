cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_large.py -x -q -m gpu > $O/pytest_large.log 2>&1; grep -E "passed|failed" $O/pytest_large.log | tail -1
for env in "X=1" "DNE_NSUB=1" "DNE_LFC_COLS_MAX=0"; do echo "== $env"; env $env timeout 600 python tools/ga_bench.py --large 2>&1 | tail -3 | cut -c1-150; done
