#!/bin/bash
# three launches per window and lock-step below 500 pairs -- the policy head, the emulator and the renderer in ONE launch
# behind the sub-slice fc (DNE_SUB_RENDER_FUSED=1: k_tail_step<.., true>) -- against four (k_tail_step + k_env_render)
set -u
TAG=${1:-r06s}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
DNE_SUB_RENDER_FUSED=1 DNE_TEST_VARIANTS=1 timeout 300 python -m pytest tests/test_gpu_edges.py -x -q -k "test_every_step_kernel_variant and (knobs9- or knobs10 or knobs11)" 2>&1 | tail -1
timeout 900 python tools/shard_ab.py --worlds 8,4 --gens 6 --warmup 3 --verify 1 "table" "table DNE_SUB_RENDER_FUSED=1" "table" "table DNE_SUB_RENDER_FUSED=1" > $O/shard_ab.jsonl 2> $O/shard_ab.err
python - <<PY
import json
for l in open("$O/shard_ab.jsonl"):
    d = json.loads(l); print(d["world"], "%-40s" % d["setting"], d["ms_per_generation"], d["rank0_ms"], d["theta_sha"], d.get("theta_matches_one_rank_evaluation"))
PY
timeout 600 python tools/ab_inproc.py "X=0" "DNE_SUB_RENDER_FUSED=1" --rounds 2 --gens 6 --skip alone,lockstep > $O/ab.jsonl 2> $O/ab.err
tail -1 $O/ab.jsonl | python -c "
import json,sys
for k,v in json.loads(sys.stdin.read())['summary'].items(): print('2500 pairs', k, 'gen', v['gen_ms'], v['theta_sha'][0][:8])"
