mkdir -p gpurun_out/r02c
for v in "0 --torch-first" "0 " "1 --torch-first" "1 "; do
  set -- $v
  DNE_STAGED_COPY=$1 timeout 300 python tools/upload_soak.py $2 --iters 30 > gpurun_out/r02c/soak_$1_${2:-notorch}.out 2> gpurun_out/r02c/soak_$1_${2:-notorch}.err
  echo "staged=$1 $2 rc=$?" >> gpurun_out/r02c/summary.txt
  tail -n 2 gpurun_out/r02c/soak_$1_${2:-notorch}.err
done
cat gpurun_out/r02c/summary.txt
