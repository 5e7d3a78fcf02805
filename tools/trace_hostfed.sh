#!/bin/bash
# tools/trace_hostfed.sh -- ON THE GPU BOX: kernel + memory-copy trace (no counters) of the host-fed entry points
# (tools/bench_hostfed.py: one synchronous call per pair, then a 32-pair batch, from pageable numpy memory -> the staged
# path) -> gpurun_out/prof_hostfed/, summarised by tools/summarize_hostfed.py into gpurun_out/<tag>_hostfed_trace.md
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_hostfed
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT -o trace -- python3 $ROOT/tools/bench_hostfed.py > $OUT/trace.log 2>&1
echo "trace exit=$? $(tail -2 $OUT/trace.log | tr '\n' ' ')"
ls $OUT
python3 $ROOT/tools/summarize_hostfed.py $OUT $ROOT/gpurun_out/${TAG}_hostfed_trace.md > /dev/null && head -30 $ROOT/gpurun_out/${TAG}_hostfed_trace.md
