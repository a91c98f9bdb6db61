cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02t
for v in "" "--sort-idx" "--idx-range 16000000" "--idx-range 64000000 --sort-idx" "--idx-range 2000000"; do
  echo "== default nsub: $v" >> gpurun_out/r02t/kbench.log
  timeout 300 python tools/kbench.py --tslimit 24 --reps 2 $v >> gpurun_out/r02t/kbench.log 2>&1
  echo "== NSUB=1: $v" >> gpurun_out/r02t/kbench.log
  DNE_NSUB=1 timeout 300 python tools/kbench.py --tslimit 24 --reps 2 $v >> gpurun_out/r02t/kbench.log 2>&1
done
cat gpurun_out/r02t/kbench.log
