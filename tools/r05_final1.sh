#!/bin/bash
# round 5: the suite once on the final build, then the round's profile set
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s -k "pool_workers" > $O/r05_pool_placement.txt 2>&1; grep "pool placement\|passed\|failed" $O/r05_pool_placement.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r05_pytest_final.txt 2>&1; rc=$?
tail -5 $O/r05_pytest_final.txt
[ $rc -eq 0 ] || exit $rc
bash tools/make_profiles.sh r05 2>&1 | tail -40
