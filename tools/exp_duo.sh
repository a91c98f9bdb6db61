cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02u
timeout 900 python -m pytest tests/test_gpu_edges.py -x -q -m gpu -k "variant" > gpurun_out/r02u/pytest_variants.log 2>&1; tail -5 gpurun_out/r02u/pytest_variants.log
L=gpurun_out/r02u/kbench.log
for v in "" "--sort-idx"; do
  for env in "DNE_FC_DUO=0" "DNE_FC_DUO=1" "DNE_FC_DUO=1 DNE_FC_RB=2" "DNE_FC_DUO=0 DNE_NSUB=1" "DNE_FC_DUO=1 DNE_NSUB=1" "DNE_FC_DUO=1 DNE_NSUB=1 DNE_FC_RB=2"; do
    echo "== $env : $v" >> $L
    env $env timeout 300 python tools/kbench.py --tslimit 24 --reps 2 $v 2>&1 | grep rep >> $L
  done
done
cat $L
