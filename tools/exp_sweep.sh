cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03j; mkdir -p $O
run() { env $1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err; echo "$1: $(tail -1 $O/b.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2))")"; }
run "X=0"
run "DNE_DUO_SOLO_BELOW=1200"
run "DNE_DUO_SOLO_BELOW=1900"
run "DNE_FC_DUO_MIN=600"
run "DNE_FC_GRID=384"
run "DNE_FC_GRID=640"
run "DNE_NSUB=4"
run "DNE_SPEC_MAX=6"
run "DNE_SPEC_MAX=12"
run "DNE_SPEC_BANDS=4"
run "DNE_FC_TAIL_MAX=64"
run "DNE_FC_TAIL_MAX=128"
run "DNE_TAIL_FUSED_MAX=100"
run "X=1"
