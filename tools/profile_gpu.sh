#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for bench.py.
# Usage: tools/profile_gpu.sh <tag> [bench args...]
# Outputs under gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
echo "== kernel trace" 
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
echo "trace exit=$?"
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$n -o pmc -- python3 $ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $OUT/pmc_$n.log 2>&1
  echo "pmc [$grp] exit=$?"
done
find $OUT -name "*.csv" | head -40
# calibration of FETCH_SIZE / WRITE_SIZE on known byte counts in the kernel's access shapes
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_calib_$c -o pmc -- $ROOT/tools/ubench/hbm_calib > $OUT/calib_$c.log 2>&1
  echo "calib [$c] exit=$?"
done
