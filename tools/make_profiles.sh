#!/bin/bash
# Round profile set, run ON THE GPU BOX (gpurun): per-configuration rocprofv3 summaries (kernel trace +
# separate PMC passes, tools/profile_cfg.sh) for BASELINE configs 2, 3, 5 and the reference's default
# parameters (16/64), a bench.py line, and the kernel-trace stats of bench.py itself.
#   tools/make_profiles.sh <round tag, e.g. r02>
# Output: gpurun_out/<tag>_*.md (+ raw CSVs under gpurun_out/prof_<tag>_*); tools/collect_profiles.py then
# copies the summaries into profiles/ and writes profiles/traffic.json with the source digest.
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
tools/profile_cfg.sh ${TAG}_c3 1920 1080 9 64 1024     # the bench's own shape (counter passes sample the disparity kernels only: profile_cfg.sh)
PMC_SQ="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM" tools/profile_cfg.sh ${TAG}_c3w 1920 1080 9 64 1024
tools/profile_cfg.sh ${TAG}_c5 3840 2160 15 128 16
PMC_SQ="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM" tools/profile_cfg.sh ${TAG}_c5w 3840 2160 15 128 16
tools/profile_cfg.sh ${TAG}_c2 640 480 7 32 128
tools/profile_cfg.sh ${TAG}_b16 1920 1080 16 64 64
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_${TAG}_bench
mkdir -p $OUT
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ROOT/bench.py --headline-only --steps 10 --warmup 2 > $OUT/trace.log 2>&1)
echo "bench trace exit=$?"
for f in $OUT/trace/*_kernel_trace.csv; do [ -f "$f" ] && (head -1 "$f"; grep -E "k_colwave|k_disparity" "$f") > "$f.tmp" && mv "$f.tmp" "$f"; done
rm -f $OUT/trace/*agent_info.csv $OUT/trace/*domain_stats.csv
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_line.err
echo "bench exit=$?"; tail -c 600 gpurun_out/${TAG}_bench_line.json
du -sh gpurun_out
