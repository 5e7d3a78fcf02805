#!/bin/bash
# ONE unfiltered counter pass over tools/bench_cfg.py at 1024 resident 1080p pairs: the run that aborted in round 2
# and whose output was not kept (VERDICT r2 "weak" 2).  Diagnostic only -- tools/profile_cfg.sh samples the disparity
# kernels alone (--kernel-include-regex) and is what every profile under profiles/ comes from.  Keeps the complete
# log, the exit status, memory before / after, and the tail of whatever CSVs were written.
#   tools/pmc_unfiltered_once.sh [NFRAMES]        (run it alone in its own gpurun call, once)
set -u
N=${1:-1024}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_pmc_unfiltered_$N
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ulimit -c 0
(date; free -g; rocm-smi --showmeminfo vram) > $OUT/before.txt 2>&1
timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o pmc -- python3 $ROOT/tools/bench_cfg.py 1920 1080 9 64 $N > $OUT/run.log 2>&1
echo "exit=$?" | tee $OUT/exit.txt
(date; free -g; rocm-smi --showmeminfo vram) > $OUT/after.txt 2>&1
for f in $OUT/pmc/*_counter_collection.csv $OUT/pmc/*_kernel_trace.csv; do
  [ -f "$f" ] || continue
  echo "$(wc -l < "$f") lines in $(basename $f)" | tee -a $OUT/exit.txt
  (head -1 "$f"; tail -12 "$f") | cut -c1-400 > "$f.tail"
  rm -f "$f"
done
rm -f $OUT/pmc/*agent_info.csv
echo "--- last lines of the log"
tail -c 2500 $OUT/run.log
