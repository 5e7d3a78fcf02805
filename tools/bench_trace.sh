#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py's headline launches alone (12 launches of 1024 pairs) + a fresh bench line.
#   tools/bench_trace.sh <tag>     -> gpurun_out/prof_<tag>_bench/trace/, gpurun_out/<tag>_bench_line.json
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_${TAG}_bench
mkdir -p $OUT
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ROOT/bench.py --headline-only --steps 10 --warmup 2 > $OUT/trace.log 2>&1)
echo "bench trace exit=$?"
for f in $OUT/trace/*_kernel_trace.csv; do [ -f "$f" ] && (head -1 "$f"; grep -E "k_colwave|k_disparity" "$f") > "$f.tmp" && mv "$f.tmp" "$f"; done
rm -f $OUT/trace/*agent_info.csv $OUT/trace/*domain_stats.csv
cd $ROOT
timeout -k 10 400 python3 bench.py --steps 20 --warmup 3 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_line.err
echo "bench exit=$?"
grep -E "k_colwave" $OUT/trace/trace_kernel_stats.csv | cut -c1-40,100-220
