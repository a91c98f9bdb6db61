cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03c; mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20.json 2> $O/bench20.err; echo "pop5000: $(tail -1 $O/bench20.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
for p in 2500 1250 624; do timeout 600 python bench.py --no-cpu-baseline --pop $p --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($p, d['value'], d['ms_per_step'])"; done
DNE_SPEC_MAX=0 timeout 600 python bench.py --no-cpu-baseline --pop 624 --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('624 nospec', d['value'], d['ms_per_step'])"
