#!/bin/bash
# Round-5 soak of the stress generators on the round's final library (fresh seeds), progress line per tool.
# usage (GPU box): tools/soak_r05.sh > gpurun_out/r05_soak.txt
cd "$(dirname "$0")/.."
set -o pipefail
t0=$(date +%s)
for job in "stress_vs_generic.py ${SOAK_X:-1}0000 5101" "stress_dev_batch.py ${SOAK_X:-1}000 5102" "stress_boundary.py ${SOAK_X:-1}500 5103" "stress_hostcopy.py ${SOAK_X:-1}000 5104"; do
  echo "== $job ($(( $(date +%s) - t0 )) s)"
  timeout -k 10 1000 python tools/$job 2>&1 | tail -4 || echo "FAILED: $job"
done
echo "== done ($(( $(date +%s) - t0 )) s)"
