cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03i; mkdir -p $O
for env in "DNE_DUO_HEAD_FUSED=0 DNE_OUT_LDS_KB=64" "DNE_DUO_HEAD_FUSED=0" "DNE_DUO_HEAD_FUSED=1" "DNE_DUO_HEAD_FUSED=0 DNE_OUT_LDS_KB=64" "DNE_DUO_HEAD_FUSED=0" "DNE_DUO_HEAD_FUSED=1" "DNE_SPEC_MAX=0" "DNE_FC_DUO=0"; do
env $env timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20.json 2> $O/bench20.err; echo "$env: $(tail -1 $O/bench20.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
done
