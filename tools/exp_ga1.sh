cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_edges.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu --deselect tests/test_gpu_fullsize.py::test_full_generation_bit_exact > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1
for env in "DNE_FC_DUO_GA=0" "DNE_FC_DUO_GA=1" "DNE_FC_DUO_GA=1 DNE_DUO_SOLO_BELOW=0"; do echo "== $env"; env $env timeout 300 python tools/ga_bench.py 2>&1 | cut -c1-120; done | tee $O/ga.log
