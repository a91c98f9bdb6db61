#!/bin/bash
# what the driver runs at round end: the GPU suite, smoke(), the bench with the driver's flags -- on the final tree
TAG=${1:-r05s}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -16 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json,sys
d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','higher_is_better','scaling','vs_baseline','dtype')})
print(d['roofline']['bound'], d['roofline']['frac'], d['roofline']['frac_counter'], d['roofline']['traffic_regime'][:40]); print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
print({k:(v.get('value'), v.get('error')) for k,v in d['extra'].items()})"
