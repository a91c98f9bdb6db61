#!/bin/bash
TAG=${1:-r04v}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
for cfg in "X=0" "DNE_LFC_PAD=1" "DNE_LFC_PAD=2" "DNE_LFC_PAD=1 DNE_FC_GRID=256"; do
  env $cfg timeout 300 python tools/ga_bench.py --large > "$O/gal.$cfg.jsonl" 2> "$O/gal.$cfg.err"
  echo "$cfg: $(tail -1 "$O/gal.$cfg.jsonl" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']))") $(grep -c . "$O/gal.$cfg.jsonl")"
done
timeout 300 python -m pytest tests/test_gpu_large.py -x -q > $O/pytest_large.log 2>&1; tail -2 $O/pytest_large.log
