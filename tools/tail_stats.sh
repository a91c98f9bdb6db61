#!/bin/bash
# Per-kernel durations of a tail lock-step under rocprofv3 (kernel trace + stats), one run per active-pair count:
#   bash tools/tail_stats.sh <tag> 1 8 24    ->  gpurun_out/<tag>/tail_stats_<pairs>.csv (Name, Calls, AverageNs ...)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for p in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ts_$p" -o t -- python "$R/tools/tail_bench.py" $p --steps 208 > "$O/tail_stats_$p.log" 2>&1
  cp "$(find "$O/ts_$p" -name '*kernel_stats.csv' | head -1)" "$O/tail_stats_$p.csv" 2>/dev/null
  rm -rf "$O/ts_$p"
  echo "== $p pairs"; grep -v "ref\|bn_\|rocclr\|reset\|iota\|ref_to" "$O/tail_stats_$p.csv" | cut -d, -f1-4 | cut -c1-150 | head -12
done
