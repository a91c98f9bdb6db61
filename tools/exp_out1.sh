cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_edges.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1
for env in "X=1"; do
  env $env timeout 300 python tools/kbench.py --tslimit 24 --reps 4 2>&1 | grep rep | tail -3 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k: round(v,3) for k,v in d['per_step_ms'].items()}, round(d['step_wall_ms'],3))"
done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20.json 2> $O/bench20.err; echo "pop5000: $(tail -1 $O/bench20.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_counter'])")"
