# Runs the driver's round-end command several times on one box; every run's stdout/stderr is kept.
out=gpurun_out/${1:-r02a}; n=${2:-3}
mkdir -p $out; rm -f $out/summary.txt
for i in $(seq 1 $n); do
  s=$(date +%s.%N)
  python3 bench.py --gpus 1 --steps 20 --warmup 5 ${BENCH_EXTRA:---no-cpu-baseline} > $out/run$i.out 2> $out/run$i.err
  echo "run$i rc=$? wall=$(echo "$(date +%s.%N) - $s" | bc)" >> $out/summary.txt
done
cat $out/summary.txt
for i in $(seq 1 $n); do echo "--- run$i"; tail -n 5 $out/run$i.err; cut -c1-400 $out/run$i.out; done
