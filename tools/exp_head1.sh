cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_edges.py -x -q -m gpu -k "variant" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1
for env in "DNE_DUO_HEAD_FUSED=0" "DNE_DUO_HEAD_FUSED=1"; do
  env $env timeout 300 python tools/kbench.py --tslimit 24 --reps 4 2>&1 | grep rep | tail -2 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$env', {k: round(v,3) for k,v in d['per_step_ms'].items()}, round(d['step_wall_ms'],3))"
done
for env in "DNE_DUO_HEAD_FUSED=0" "DNE_DUO_HEAD_FUSED=1" "DNE_DUO_HEAD_FUSED=0" "DNE_DUO_HEAD_FUSED=1"; do
env $env timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20.json 2> $O/bench20.err; echo "$env: $(tail -1 $O/bench20.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
done
