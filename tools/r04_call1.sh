#!/bin/bash
# round 4, first GPU call: host facts + CPU-baseline sweep, the new full-size parity tests, PMC on the bench's own launch mix
TAG=${1:-r04a}
O=gpurun_out/$TAG; mkdir -p $O
python tools/hostinfo.py > $O/host.json 2>&1; cat $O/host.json
nproc > $O/nproc.txt; cat /sys/fs/cgroup/cpu.max >> $O/nproc.txt 2>&1; lscpu | head -20 >> $O/nproc.txt
timeout 600 python bench.py --steps 2 --warmup 1 --extra config1 > $O/bench_small.json 2> $O/bench_small.err; echo "bench rc=$?"
python - <<'PY' $O
import json,sys
d=json.loads([l for l in open(sys.argv[1]+'/bench_small.json') if l.startswith('{')][-1])
print('value',d['value'],'ms',d['ms_per_step']); c=d['cpu_baseline']; print('cpu',c['value'],c['cores'],c['cpus_delivered'])
for r in c['sweep']: print('  ',r)
print(json.dumps(d['roofline'].get('floors')), d['roofline']['frac_counter'], d['roofline']['traffic_regime'])
PY
timeout 1500 python -m pytest tests/test_gpu_fullsize_configs.py tests/test_gpu_fullsize.py::test_full_generation_bit_exact -x -q --durations=8 > $O/pytest_new.log 2>&1; echo "pytest rc=$?"
tail -25 $O/pytest_new.log
timeout 900 bash tools/collect_pmc_bench_mix.sh $TAG > $O/pmc_mix.log 2>&1; echo "pmc rc=$?"; tail -12 $O/pmc_mix.log
