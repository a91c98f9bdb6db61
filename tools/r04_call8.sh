#!/bin/bash
# whole GPU suite on the new defaults (k_fc_sub in the mid range of ES and GA) + GA profile + the driver's command
TAG=${1:-r04h}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -22 $O/pytest_gpu.log
timeout 300 python tools/ga_lockstep_profile.py > $O/ga_prof.json 2> $O/ga_prof.err
python - "$O/ga_prof.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); g=d["generation_1"]
print("GA gen1 %.1f ms %.0f steps/s" % (g["wall_ms"], g["steps_per_s"]), d["lock_step_us_at_width"])
PY
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc=$?"
python - "$O/bench_driver_cmd.json" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('value',d['value'],'ms',d['ms_per_step'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'ratio', d['cpu_baseline'].get('gpu_over_cpu'))
for k,v in d['extra'].items(): print(' ',k, v.get('value'), v.get('error'), (v.get('cpu_baseline') or {}).get('value'), [ (x['chain'], round(x['cold']['steps_per_s']), round(x['rebuild_ms'],1)) for x in v.get('deep_chains',[])])
PY
