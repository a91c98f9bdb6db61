cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03g; mkdir -p $O
timeout 600 python tools/ga_bench.py --large > $O/ga_large_bench.jsonl 2> $O/ga_large_bench.err; cat $O/ga_large_bench.jsonl | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rm -rf $O/st; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o g -- python $GRAFT_REPO_ROOT/tools/ga_bench.py --large > /dev/null 2>&1
f=$(find $O/st -name "*kernel_stats.csv" | head -1); cp $f $O/ga_large_kernel_stats.csv; head -12 $f | cut -c1-170
find $O/st -name "*kernel_trace.csv" -delete
