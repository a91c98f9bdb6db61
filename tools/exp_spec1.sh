cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py::test_full_generation_bit_exact > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
for env in "DNE_SPEC_MAX=0" "DNE_SPEC_MAX=8 DNE_SPEC_CONV1=0" "DNE_SPEC_MAX=8" "DNE_SPEC_MAX=16"; do
  echo "== $env"; env $env timeout 300 python tools/tail_bench.py 1,2,4,8 2>&1 | tail -1
done | tee $O/tail.log
