cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_large.py tests/test_gpu_parity.py tests/test_gpu_edges.py -x -q -m gpu > $O/pytest_large.log 2>&1; tail -4 $O/pytest_large.log
timeout 600 python tools/ga_bench.py --large 2>&1 | tail -3 | tee $O/ga_large.jsonl
