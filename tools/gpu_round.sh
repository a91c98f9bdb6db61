mkdir -p gpurun_out/r02b
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r02b/pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r02b/summary.txt
tail -25 gpurun_out/r02b/pytest.log
BENCH_EXTRA=" " bash tools/repro_bench.sh r02b 2
