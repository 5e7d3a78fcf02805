#!/bin/bash
# Per-configuration profile on the GPU box: kernel trace + three separate PMC passes over
# tools/bench_cfg.py (device-resident batch of one configuration), summarised by
# tools/summarize_cfg.py into profiles/.
#   tools/profile_cfg.sh <tag> W H BLOCK MAXDISP NFRAMES
# Output: gpurun_out/prof_<tag>/{trace,pmc_sq,pmc_fetch,pmc_write}/
# The counter passes sample the disparity kernels ONLY (--kernel-include-regex, always): bench_cfg.py builds its frames
# with ~40 torch elementwise launches per pair, and counter collection over all of them is what crashed the profiled
# process at 1024 resident pairs in rounds 2 and 3 (a host-side SIGSEGV inside the profiler's dispatch interception,
# DESIGN.md section 6; profiles/r03_pmc_unfiltered_1024pairs_FAIL.log).  There is no switch to run unfiltered.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
FARGS=(--kernel-include-regex "k_colwave|k_disparity|k_gray|k_sad_windows")
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ROOT/tools/bench_cfg.py "$@" > $OUT/trace.log 2>&1
echo "[$TAG] trace exit=$? $(tail -1 $OUT/trace.log)"
timeout -k 10 420 rocprofv3 --kernel-trace "${FARGS[@]}" --pmc ${PMC_SQ:-SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES} GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o pmc -- python3 $ROOT/tools/bench_cfg.py "$@" > $OUT/pmc_sq.log 2>&1
echo "[$TAG] pmc sq exit=$? $(grep -c . $OUT/pmc_sq.log) log lines"
timeout -k 10 420 rocprofv3 --kernel-trace "${FARGS[@]}" --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $ROOT/tools/bench_cfg.py "$@" > $OUT/pmc_fetch.log 2>&1
echo "[$TAG] pmc fetch exit=$?"
timeout -k 10 420 rocprofv3 --kernel-trace "${FARGS[@]}" --pmc GRBM_GUI_ACTIVE WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $ROOT/tools/bench_cfg.py "$@" > $OUT/pmc_write.log 2>&1
echo "[$TAG] pmc write exit=$?"
# keep only the disparity kernels' rows of the big CSVs (torch's generator kernels dominate them):
# gpurun copies back at most 64 MiB
for f in $OUT/*/*_kernel_trace.csv $OUT/*/*_counter_collection.csv; do
  [ -f "$f" ] || continue
  (head -1 "$f"; grep -E "k_colwave|k_disparity|k_fcol|k_mcol|k_read|k_write" "$f") > "$f.tmp" && mv "$f.tmp" "$f"
done
rm -f $OUT/*/*agent_info.csv $OUT/*/*domain_stats.csv
python3 $ROOT/tools/summarize_cfg.py $TAG $ROOT/gpurun_out/$TAG.md "$@" > /dev/null && tail -12 $ROOT/gpurun_out/$TAG.md
