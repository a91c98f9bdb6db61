#!/bin/bash
# reference-pass kernels: same-box A/B of the committed library against the working tree (per-kernel durations of ONE chunk of 5000 members)
TAG=${1:-r04ref}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_PYTEST" ]; then timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/pytest.log | tail -2; fi
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
