cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/st; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o g -- python $GRAFT_REPO_ROOT/tools/ga_bench.py --large > /dev/null 2>&1
f=$(find $O/st -name "*kernel_stats.csv" | head -1); head -9 $f | cut -c1-200
find $O/st -name "*kernel_trace.csv" -delete
