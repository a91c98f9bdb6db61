#!/bin/bash
# round 4, second GPU call: VALU issue-rate micro-benchmark, the W = 4 variants' bit-exactness, same-process A/B of k_fc_duo's W
TAG=${1:-r04b}
O=gpurun_out/$TAG; mkdir -p $O
./tools/micro/valu_rate > $O/valu_rate.jsonl 2> $O/valu_rate.err; echo "valu rc=$?"; cat $O/valu_rate.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('%-90s w/simd %d  cyc/instr(wave) %.2f  cyc/instr/simd %.2f' % (d['kind'][:90], d['waves_per_simd'], d['cycles_at_2p4GHz_per_wave_instr'], d['cycles_per_instr_per_simd']))"
timeout 900 python -m pytest tests/test_gpu_edges.py -x -q -k "variant" > $O/pytest_variants.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_variants.log
timeout 900 python tools/ab_inproc.py "X=0" "DNE_DUO_W=4" "DNE_DUO_W=4 DNE_DUO_SYNC=2" "DNE_DUO_W=4 DNE_DUO_GRID=768" "DNE_DUO_W=4 DNE_FC_PRIO=1" > $O/ab1.jsonl 2> $O/ab1.err; echo "ab rc=$?"; cat $O/ab1.jsonl; tail -3 $O/ab1.err
