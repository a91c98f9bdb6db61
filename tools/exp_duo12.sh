cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_edges.py -x -q -m gpu -k "variant" > $O/pytest_variants.log 2>&1; tail -3 $O/pytest_variants.log
for env in "DNE_DUO_SOLO_BELOW=1500" "DNE_DUO_SOLO_BELOW=0" "DNE_DUO_SOLO_BELOW=1900"; do
env $env timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20.json 2> $O/bench20.err; echo "$env: $(tail -1 $O/bench20.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])")"
done
