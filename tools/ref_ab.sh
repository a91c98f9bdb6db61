#!/bin/bash
# reference-pass kernels: parity, same-box A/B against a baseline library, per-kernel durations (ONE chunk of 5000 members under rocprofv3),
# phase clock.  The baseline is whatever was built into deep-neuroevolution_amd/csrc/libdne_hip_base.so beforehand, e.g. the last commit:
#   git stash; make -C deep-neuroevolution_amd/csrc libdne_hip.so; cp deep-neuroevolution_amd/csrc/libdne_hip.so deep-neuroevolution_amd/csrc/libdne_hip_base.so; git stash pop; make -C deep-neuroevolution_amd/csrc libdne_hip.so clock
#   gpurun --timeout 1200 -- 'bash tools/ref_ab.sh <tag>'
TAG=${1:-r04ref}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed|error" $O/pytest.log | tail -2
for lib in base new; do
  if [ $lib = base ]; then export DNE_LIB_PATH=$PWD/deep-neuroevolution_amd/csrc/libdne_hip_base.so; else unset DNE_LIB_PATH; fi
  python tools/ref_bench.py > $O/ref_$lib.json 2> $O/ref_$lib.err; cat $O/ref_$lib.json
  REF_CHUNK=5000 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$lib -o x -- python tools/ref_bench.py > $O/ref1_$lib.json 2> $O/ref1_$lib.err
  f=$(find $O/prof_$lib -name '*kernel_stats.csv' | head -1); cp $f $O/ref_one_chunk_kernel_stats_$lib.csv; rm -rf $O/prof_$lib
  python - $O/ref_one_chunk_kernel_stats_$lib.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    print("  %-50s calls %s avg %.3f ms" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done
DNE_LIB_PATH=$PWD/deep-neuroevolution_amd/csrc/libdne_hip_clock.so REF_CHUNK=5000 python tools/ref_phase_clock.py > $O/ref_phase_clock.json 2> $O/pc.err; python -c "
import json; d=json.load(open('$O/ref_phase_clock.json'))
for k,v in d.items(): print(k, v['us_per_workgroup'], v['phases_us'])"
