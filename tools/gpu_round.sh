# one GPU-box round: the GPU test suite, then whatever extra commands were given (each logged under gpurun_out/<tag>/)
TAG=${1:-r02}; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 ${PYTEST_EXTRA:-} > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/summary.txt
tail -15 $O/pytest.log
i=0
for cmd in "$@"; do
  i=$((i+1))
  echo "== $cmd" | tee -a $O/summary.txt
  timeout 900 bash -c "$cmd" > $O/cmd$i.out 2> $O/cmd$i.err; echo "rc=$?" | tee -a $O/summary.txt
  tail -c 3000 $O/cmd$i.out; tail -n 4 $O/cmd$i.err
done
