cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02z; mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20_duo.json 2> $O/bench20_duo.err; tail -1 $O/bench20_duo.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'], d.get('roofline_ref_pass'), d['cpu_baseline'])"
DNE_FC_DUO=0 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20_fc2.json 2> $O/bench20_fc2.err; tail -1 $O/bench20_fc2.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
